// Multi-head graph attention forward pass — same classes as the reference's gat.hpp (GATLayer :25-41, GAT :50-113).
// Per head (computeSelfAttentionHead, gat.hpp:83-104):
//     A = buffers[i] * W_j            dense GEMM          -> hnh_gemm_f64 (fp64 MFMA: the path's one dense contraction)
//     B = A;  de_shift(&B, nullptr, k_spmmA)
//     e = SDDMM(A, B)  (S = 1)        algorithm(k_sddmmA, replicate)
//     e = LeakyReLU(e)                hnh_leaky_relu_f64 on the value vector
//     A = 0;  A = SpMM(e, B)          algorithm(k_spmmA, no re-replication: FusedMM by replication reuse)
//     buffers[i+1][:, j*w : (j+1)*w] = ReLU(A)            -> hnh_relu_store_cols_f64
// The reference leaves the weights zero and `leaky_relu_alpha` uninitialised (gat.hpp:55,78):
// it is a timing skeleton.  Here alpha defaults to 0.2 and weights are settable; like the reference, the product
// X * W uses each rank's LOCAL column slice of X (gat.hpp:88), so results are only meaningful for schedules that
// do not split R (1.5D dense shift), which is what the parity tests use.
#pragma once
#include "distributed_sparse.hpp"
#include <cstdlib>

class GATLayer {
public:
    int input_features, features_per_head, num_heads;
    std::vector<DenseMatrix> wMats;
    VectorXd a1, a2;
    GATLayer(int input_features, int features_per_head, int num_heads)
        : input_features(input_features), features_per_head(features_per_head), num_heads(num_heads) {}
};

class GAT {
public:
    Distributed_Sparse* d_ops;
    std::vector<GATLayer> layers;
    std::vector<DenseMatrix> buffers;
    double leaky_relu_alpha = 0.2;

    GAT(std::vector<GATLayer>& l_input, Distributed_Sparse* d_ops) {
        if (l_input.empty()) hnh::fatal("Error, a GAT needs at least one layer!");
        this->d_ops = d_ops;
        world_ = d_ops->world;  // not owned by the operator: benchmark_dist.cpp:166 deletes d_ops while its GAT is still alive
        layers = l_input;
        d_ops->setRValue(layers[0].input_features);
        buffers.push_back(d_ops->like_B_matrix(0.0));
        for (size_t i = 0; i < layers.size(); i++) {
            if (i > 0 && layers[i].input_features != layers[i - 1].num_heads * layers[i - 1].features_per_head)
                hnh::fatal("Error, GAT layer input width does not match the previous layer's output width!");
            d_ops->setRValue(layers[i].features_per_head * layers[i].num_heads);
            buffers.push_back(d_ops->like_A_matrix(0.0));
            d_ops->setRValue(layers[i].features_per_head);
            for (int j = 0; j < layers[i].num_heads; j++)
                layers[i].wMats.push_back(DenseMatrix::Constant(buffers[i].cols(), d_ops->localAcols, 0.0));
        }
    }

    ~GAT() {  // never touches d_ops: the reference's harness has deleted it by now (benchmark_dist.cpp:166 vs the unique_ptr's scope)
        hnh::World* w = world_;
        if (w == nullptr) return;
        w->sync_all_nothrow();  // the events below may still be waited on
        for (void* e : {ev_input, ev_gemm[0], ev_gemm[1], ev_head[0], ev_head[1]})
            if (e) w->event_destroy(e);
    }
    GAT(const GAT&) = delete;
    GAT& operator=(const GAT&) = delete;

    // Computes the j'th self-attention head of the i'th layer (gat.hpp:83-104)
    void computeSelfAttentionHead(int i, int j) {
        DenseMatrix A;
        head_product(i, j, A, HNH_STREAM_COMPUTE);
        head_attention(i, j, A);
    }

    // The reference runs the heads one after the other (gat.hpp:106-112).  Here a layer is a two-stage pipeline: the product
    // X * W of head j + 1 (MFMA bound, HNH_STREAM_AUX) runs beside the attention pass of head j (HBM bound, the compute stream),
    // between two product buffers; the layers' outputs are the reference's, bit for bit (same kernels, same operands).
    // HNH_GAT_SERIAL=1: the reference's order on one stream (A/B measurements).
    void forwardPass() {
        if (std::getenv("HNH_GAT_SERIAL") != nullptr) {
            for (size_t i = 0; i < layers.size(); i++)
                for (int j = 0; j < layers[i].num_heads; j++) computeSelfAttentionHead((int)i, j);
            return;
        }
        hnh::World* w = d_ops->world;
        if (!ev_input)
            for (void** e : {&ev_input, &ev_gemm[0], &ev_gemm[1], &ev_head[0], &ev_head[1]}) *e = w->event_create();
        for (size_t i = 0; i < layers.size(); i++) {
            const int H = layers[i].num_heads;
            // (allocated before the mark below: the allocator orders a recycled block behind its last use on the compute and
            // communication streams, and the auxiliary stream inherits that through the mark)
            for (int b = 0; b < (H > 1 ? 2 : 1); b++) shape_product((int)i, product[b]);
            // the layer's input is complete, and both product buffers are free, once the compute stream gets here
            w->event_record(ev_input, HNH_STREAM_COMPUTE);
            w->event_wait(ev_input, HNH_STREAM_AUX);
            head_product((int)i, 0, product[0], HNH_STREAM_AUX);
            w->event_record(ev_gemm[0], HNH_STREAM_AUX);
            for (int j = 0; j < H; j++) {
                w->event_wait(ev_gemm[j % 2], HNH_STREAM_COMPUTE);
                if (j + 1 < H) {
                    // product[(j + 1) % 2] was last read by head j - 1, which the compute stream has been given already
                    if (j >= 1) w->event_wait(ev_head[(j - 1) % 2], HNH_STREAM_AUX);
                    head_product((int)i, j + 1, product[(j + 1) % 2], HNH_STREAM_AUX);
                    w->event_record(ev_gemm[(j + 1) % 2], HNH_STREAM_AUX);
                }
                head_attention((int)i, j, product[j % 2]);
                w->event_record(ev_head[j % 2], HNH_STREAM_COMPUTE);
            }
        }
        // every product was awaited by the compute stream: nothing is left on the auxiliary stream
    }

private:
    hnh::World* world_ = nullptr;
    DenseMatrix product[2];  // X * W_j of the head in flight and of the next one
    void* ev_input = nullptr;
    void* ev_gemm[2] = {nullptr, nullptr};
    void* ev_head[2] = {nullptr, nullptr};

    void shape_product(int i, DenseMatrix& A) {
        const int64_t rows = buffers[i].rows(), cols = layers[i].wMats[0].cols();
        if (A.rows() != rows || A.cols() != cols) A = DenseMatrix(rows, cols);
    }

    // A = buffers[i] * W_j (gat.hpp:88) on `stream`
    void head_product(int i, int j, DenseMatrix& A, int stream) {
        hnh::World* w = d_ops->world;
        DenseMatrix& X = buffers[i];
        DenseMatrix& W = layers[i].wMats[j];
        if (X.cols() != W.rows()) hnh::fatal("Error, GAT weight shape does not match the layer input!");
        shape_product(i, A);
        w->check(w->be->hnh_gemm_f64(w->ctx, X.rows(), W.cols(), X.cols(), X.data(), W.data(), A.data(), stream), "hnh_gemm_f64");
    }

    // the rest of the head (gat.hpp:89-101) from its product A, on the compute stream; A is consumed (the unfused route zeroes it)
    void head_attention(int i, int j, DenseMatrix& A) {
        hnh::World* w = d_ops->world;
        d_ops->setRValue(layers[i].features_per_head);
        DenseMatrix& out = buffers[i + 1];

        // Schedules with a single fused pass (1.5D dense shift, local kernel fusion, c = 1: its shifts are empty and
        // the attention matrix is not exported) do SDDMM, LeakyReLU and SpMM in ONE gather of the neighbours' rows.
        if (d_ops->c == 1) {
            // ... and the head's ReLU output (gat.hpp:101) leaves the same launch straight into its column block of the layer
            // output: H only carries partial sums between the launches of a pass
            DenseMatrix H(A.rows(), A.cols());
            hnh_fused_extras ex = {leaky_relu_alpha, 0.0, nullptr, nullptr, out.data() + (int64_t)j * A.cols(), (int64_t)out.cols()};
            if (d_ops->fusedSpMM_out(A, A, Amat, H, true, ex)) return;
        }

        VectorXd Svalues = d_ops->like_S_values(1.0);
        VectorXd sddmm_buffer = d_ops->like_S_values(1.0);
        DenseMatrix B = A;
        d_ops->de_shift(&B, nullptr, k_spmmA);

        d_ops->algorithm(A, B, Svalues, &sddmm_buffer, k_sddmmA, true);  // SDDMM phase
        A.setZero();
        w->check(w->be->hnh_leaky_relu_f64(w->ctx, sddmm_buffer.data(), leaky_relu_alpha, sddmm_buffer.size(), HNH_STREAM_COMPUTE),
                 "hnh_leaky_relu_f64");
        d_ops->algorithm(A, B, sddmm_buffer, nullptr, k_spmmA, false);   // SpMM phase, replication reused
        w->check(w->be->hnh_relu_store_cols_f64(w->ctx, out.data(), out.cols(), (int64_t)j * A.cols(), A.data(), A.rows(), A.cols(),
                                                HNH_STREAM_COMPUTE),
                 "hnh_relu_store_cols_f64");
    }
};
