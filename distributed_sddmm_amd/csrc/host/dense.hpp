// Device-resident stand-ins for the two Eigen types of the reference's API:
//   DenseMatrix = Matrix<double, Dynamic, Dynamic, RowMajor>  (common.h:13)   -> hnh::DenseMatrix
//   VectorXd                                                                   -> hnh::VectorXd
// They expose the subset of the Eigen surface the hot-path headers use (ctor(rows, cols),
// Constant, rows/cols/size/data, setZero, copy-assign, squaredNorm, cwiseProduct) but the storage lives in
// HBM: data() is a DEVICE pointer.  Copying 1 GiB operands over PCIe per call would cost as much as the
// whole fused kernel (SURVEY §7 hard part 3), so host mirrors exist only on explicit request
// (copy_from_host / copy_to_host).  All device work is enqueued on the owner's compute stream.
#pragma once
#include <cstdint>
#include <utility>
#include <vector>
#include "world.hpp"

namespace hnh {

class DeviceArray {
public:
    DeviceArray() {}
    DeviceArray(World* w, size_t bytes) : w_(w), bytes_(bytes), owned_(true) { p_ = w_->dmalloc(bytes ? bytes : 16); }
    DeviceArray(World* w, void* external, size_t bytes) : w_(w), p_(external), bytes_(bytes), owned_(false) {}
    ~DeviceArray() { reset(); }
    DeviceArray(const DeviceArray&) = delete;
    DeviceArray& operator=(const DeviceArray&) = delete;
    DeviceArray(DeviceArray&& o) noexcept { swap(o); }
    DeviceArray& operator=(DeviceArray&& o) noexcept {
        if (this != &o) { reset(); swap(o); }
        return *this;
    }
    void swap(DeviceArray& o) noexcept {
        std::swap(w_, o.w_); std::swap(p_, o.p_); std::swap(bytes_, o.bytes_); std::swap(owned_, o.owned_);
    }
    void reset() {
        if (p_ && owned_ && w_) w_->dfree(p_);  // stream-ordered: the world's pool parks it behind events
        p_ = nullptr; bytes_ = 0; owned_ = false;
    }
    void* ptr() const { return p_; }
    size_t bytes() const { return bytes_; }
    bool owned() const { return owned_; }
    World* world() const { return w_; }

private:
    World* w_ = nullptr;
    void* p_ = nullptr;
    size_t bytes_ = 0;
    bool owned_ = false;
};

class DenseMatrix {
public:
    DenseMatrix() {}
    DenseMatrix(int64_t rows, int64_t cols) : buf_(current_world(), bytes_for(rows, cols)), rows_(rows), cols_(cols) {}
    // non-owning view over device memory allocated elsewhere (e.g. a torch tensor)
    static DenseMatrix view(double* device_ptr, int64_t rows, int64_t cols) {
        DenseMatrix m;
        m.buf_ = DeviceArray(current_world(), device_ptr, bytes_for(rows, cols));
        m.rows_ = rows; m.cols_ = cols;
        return m;
    }
    static DenseMatrix Constant(int64_t rows, int64_t cols, double v) {
        DenseMatrix m(rows, cols);
        m.setConstant(v);
        return m;
    }
    DenseMatrix(const DenseMatrix& o) { *this = o; }
    DenseMatrix& operator=(const DenseMatrix& o) {
        if (this == &o) return *this;
        if (!o.buf_.ptr()) { buf_.reset(); rows_ = cols_ = 0; return *this; }
        if (o.rows_ != rows_ || o.cols_ != cols_ || !buf_.ptr()) {
            buf_ = DeviceArray(o.buf_.world(), bytes_for(o.rows_, o.cols_));
            rows_ = o.rows_; cols_ = o.cols_;
        }
        world()->copy(data(), o.data(), (size_t)size() * sizeof(double), HNH_COPY_D2D, HNH_STREAM_COMPUTE);
        return *this;
    }
    DenseMatrix(DenseMatrix&& o) noexcept { swap(o); }
    DenseMatrix& operator=(DenseMatrix&& o) noexcept {
        if (this != &o) { buf_.reset(); rows_ = cols_ = 0; swap(o); }
        return *this;
    }
    void swap(DenseMatrix& o) noexcept { buf_.swap(o.buf_); std::swap(rows_, o.rows_); std::swap(cols_, o.cols_); }

    int64_t rows() const { return rows_; }
    int64_t cols() const { return cols_; }
    int64_t size() const { return rows_ * cols_; }
    double* data() const { return static_cast<double*>(buf_.ptr()); }
    bool owns_storage() const { return buf_.owned(); }
    World* world() const { return buf_.world(); }

    void setZero() { if (size()) world()->memset0(data(), (size_t)size() * sizeof(double), HNH_STREAM_COMPUTE); }
    // the slices of Eigen's interface that code written against the reference uses on its dense matrices
    // (`tmp *= 0.0`, 15D_sparse_shift.hpp:236; `localA.middleRows(start, n)` as r- and l-value, :233,248)
    DenseMatrix& operator*=(double s) {
        if (size()) world()->check(world()->be->hnh_axpy_f64(world()->ctx, data(), data(), s - 1.0, size(), HNH_STREAM_COMPUTE), "hnh_axpy_f64");
        return *this;
    }
    DenseMatrix middleRows(int64_t start, int64_t n) const {  // a view: assigning to it writes through
        if (start < 0 || n < 0 || start + n > rows_) fatal("Error, middleRows out of range!");
        return view(data() + start * cols_, n, cols_);
    }
    void setConstant(double v) {
        if (size()) world()->check(world()->be->hnh_fill_f64(world()->ctx, data(), size(), v, HNH_STREAM_COMPUTE), "hnh_fill_f64");
    }
    void copy_from_host(const double* host) {
        world()->copy(data(), host, (size_t)size() * sizeof(double), HNH_COPY_H2D, HNH_STREAM_COMPUTE);
        world()->sync(HNH_STREAM_COMPUTE);
    }
    void copy_to_host(double* host) const {
        world()->sync_all();
        world()->copy(host, data(), (size_t)size() * sizeof(double), HNH_COPY_D2H, HNH_STREAM_COMPUTE);
        world()->sync(HNH_STREAM_COMPUTE);
    }
    std::vector<double> to_host() const {
        std::vector<double> h((size_t)size());
        if (size()) copy_to_host(h.data());
        return h;
    }
    double squaredNorm() const {  // fingerprints (scratch.cpp:45-68): host-side reduction, test utility
        double s = 0.0;
        for (double x : to_host()) s += x * x;
        return s;
    }

private:
    static size_t bytes_for(int64_t r, int64_t c) { return (size_t)r * (size_t)c * sizeof(double); }
    DeviceArray buf_;
    int64_t rows_ = 0, cols_ = 0;
};

class VectorXd {
public:
    VectorXd() {}
    explicit VectorXd(int64_t n) : m_(n, 1) {}
    static VectorXd Constant(int64_t n, double v) {
        VectorXd x(n);
        x.m_.setConstant(v);
        return x;
    }
    static VectorXd view(double* device_ptr, int64_t n) {
        VectorXd x;
        x.m_ = DenseMatrix::view(device_ptr, n, 1);
        return x;
    }
    int64_t size() const { return m_.size(); }
    double* data() const { return m_.data(); }
    World* world() const { return m_.world(); }
    void setZero() { m_.setZero(); }
    void setConstant(double v) { m_.setConstant(v); }
    void copy_from_host(const double* h) { m_.copy_from_host(h); }
    void copy_to_host(double* h) const { m_.copy_to_host(h); }
    std::vector<double> to_host() const { return m_.to_host(); }
    double squaredNorm() const { return m_.squaredNorm(); }
    VectorXd cwiseProduct(const VectorXd& o) const {
        if (o.size() != size()) fatal("Error, cwiseProduct size mismatch");
        VectorXd out(size());
        if (size())
            world()->check(world()->be->hnh_hadamard_f64(world()->ctx, out.data(), data(), o.data(), size(), HNH_STREAM_COMPUTE),
                           "hnh_hadamard_f64");
        return out;
    }

private:
    DenseMatrix m_;
};

// Double buffer for a moving dense operand (common.h:49-93).  The reference allocates `extra` on every
// algorithm() call; here the owner passes in persistent spare storage.
class BufferPair {
public:
    DenseMatrix* original;
    DenseMatrix* extra;
    int switchVal;
    BufferPair(DenseMatrix* buf, DenseMatrix* spare) : original(buf), extra(spare), switchVal(0) {
        if (extra->rows() != buf->rows() || extra->cols() != buf->cols()) *extra = DenseMatrix(buf->rows(), buf->cols());
    }
    DenseMatrix* getActive() { return switchVal == 0 ? original : extra; }
    DenseMatrix* getPassive() { return switchVal == 0 ? extra : original; }
    void swapActive() { switchVal = 1 - switchVal; }
    // After an odd number of shifts the live data is in `extra`: hand it back to the caller's matrix.
    // Owned storage is swapped in O(1); a view over external memory has to be copied (common.h:88-92).
    void sync_active() {
        if (switchVal == 1) {
            if (original->owns_storage() && extra->owns_storage()) original->swap(*extra);
            else *original = *extra;
            switchVal = 0;
        }
    }
};

}  // namespace hnh
