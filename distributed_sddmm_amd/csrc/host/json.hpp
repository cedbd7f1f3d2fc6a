// hnh::json — the small part of a JSON value type that the reference's reporting path uses.
//
// The reference returns nlohmann `json` objects from Distributed_Sparse::json_algorithm_info() / json_perf_statistics()
// (distributed_sparse.h:131-179, 245-261) and its harness indexes, nests and dumps them (benchmark_dist.cpp:144-162):
//     json j_obj;  j_obj["elapsed"] = elapsed;  j_obj["alg_info"] = d_ops->json_algorithm_info();  fout << j_obj.dump(4);
// nlohmann/json is a vendored third-party header there (26 k lines, not part of this repository); this is an independent
// value type with the same spelling for exactly those operations: null / bool / integer / floating / string / array / object
// (insertion-ordered), operator[] by key and by index, push_back, json::array(), brace initialisation of an object from
// {key, value} pairs, implicit conversions to the arithmetic types and std::string, dump() and dump(indent).
// include/compat/json.hpp puts it where `#include "json.hpp"` + `using json = nlohmann::json;` expects it.
#pragma once
#include <cstdint>
#include <cstdio>
#include <initializer_list>
#include <ostream>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace hnh {

class json {
public:
    enum class kind { null, boolean, integer, floating, string, array, object };

    json() = default;
    json(std::nullptr_t) {}
    json(bool v) : k_(kind::boolean), b_(v) {}
    template <typename T, typename std::enable_if<std::is_integral<T>::value && !std::is_same<T, bool>::value, int>::type = 0>
    json(T v) : k_(kind::integer), i_((long long)v), unsigned_(std::is_unsigned<T>::value), u_((unsigned long long)v) {}
    template <typename T, typename std::enable_if<std::is_floating_point<T>::value, int>::type = 0>
    json(T v) : k_(kind::floating), d_((double)v) {}
    json(const char* s) : k_(kind::string), s_(s) {}
    json(const std::string& s) : k_(kind::string), s_(s) {}
    template <typename T>
    json(const std::vector<T>& v) : k_(kind::array) {
        for (const T& x : v) a_.emplace_back(x);
    }
    // {{"key", value}, ...} is an object when every element is a two-element array whose first entry is a string; anything
    // else in braces is an array (nlohmann's rule)
    json(std::initializer_list<json> init) {
        bool pairs = init.size() > 0;
        for (const json& e : init) pairs = pairs && e.k_ == kind::array && e.a_.size() == 2 && e.a_[0].k_ == kind::string;
        if (pairs) {
            k_ = kind::object;
            for (const json& e : init) o_.emplace_back(e.a_[0].s_, e.a_[1]);
        } else {
            k_ = kind::array;
            a_.assign(init.begin(), init.end());
        }
    }
    static json array() {
        json j;
        j.k_ = kind::array;
        return j;
    }
    static json object() {
        json j;
        j.k_ = kind::object;
        return j;
    }

    kind type() const { return k_; }
    bool is_null() const { return k_ == kind::null; }
    bool is_object() const { return k_ == kind::object; }
    bool is_array() const { return k_ == kind::array; }
    size_t size() const { return k_ == kind::array ? a_.size() : (k_ == kind::object ? o_.size() : (k_ == kind::null ? 0 : 1)); }
    bool contains(const std::string& key) const {
        for (auto& kv : o_)
            if (kv.first == key) return true;
        return false;
    }

    // a null value becomes an object / array on first use, as in nlohmann
    json& operator[](const std::string& key) {
        if (k_ == kind::null) k_ = kind::object;
        if (k_ != kind::object) throw std::domain_error("hnh::json: operator[](key) on a value that is not an object");
        for (auto& kv : o_)
            if (kv.first == key) return kv.second;
        o_.emplace_back(key, json());
        return o_.back().second;
    }
    json& operator[](const char* key) { return (*this)[std::string(key)]; }
    const json& at(const std::string& key) const {
        for (auto& kv : o_)
            if (kv.first == key) return kv.second;
        throw std::out_of_range("hnh::json: no key " + key);
    }
    const json& operator[](const std::string& key) const { return at(key); }
    const json& operator[](const char* key) const { return at(std::string(key)); }
    template <typename I, typename std::enable_if<std::is_integral<I>::value, int>::type = 0>
    json& operator[](I index) {
        if (k_ != kind::array || (size_t)index >= a_.size()) throw std::out_of_range("hnh::json: array index");
        return a_[(size_t)index];
    }
    template <typename I, typename std::enable_if<std::is_integral<I>::value, int>::type = 0>
    const json& operator[](I index) const {
        if (k_ != kind::array || (size_t)index >= a_.size()) throw std::out_of_range("hnh::json: array index");
        return a_[(size_t)index];
    }
    void push_back(const json& v) {
        if (k_ == kind::null) k_ = kind::array;
        if (k_ != kind::array) throw std::domain_error("hnh::json: push_back on a value that is not an array");
        a_.push_back(v);
    }

    template <typename T>
    T get() const {
        if (k_ == kind::integer) return unsigned_ ? (T)u_ : (T)i_;
        if (k_ == kind::floating) return (T)d_;
        if (k_ == kind::boolean) return (T)b_;
        throw std::domain_error("hnh::json: not a number");
    }
    template <typename T, typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0>
    operator T() const {
        return get<T>();
    }
    operator std::string() const {
        if (k_ != kind::string) throw std::domain_error("hnh::json: not a string");
        return s_;
    }

    // indent < 0: one line ("{"a": 1, "b": [1, 2]}" with a space after separators, what the reference's plots read either way)
    std::string dump(int indent = -1) const {
        std::string out;
        write(out, indent, 0);
        return out;
    }

private:
    kind k_ = kind::null;
    bool b_ = false;
    long long i_ = 0;
    bool unsigned_ = false;
    unsigned long long u_ = 0;
    double d_ = 0.0;
    std::string s_;
    std::vector<json> a_;
    std::vector<std::pair<std::string, json>> o_;

    static void quote(std::string& out, const std::string& s) {
        out += '"';
        for (unsigned char c : s) {
            switch (c) {
                case '"': out += "\\\""; break;
                case '\\': out += "\\\\"; break;
                case '\n': out += "\\n"; break;
                case '\r': out += "\\r"; break;
                case '\t': out += "\\t"; break;
                default:
                    if (c < 0x20) {
                        char buf[8];
                        std::snprintf(buf, sizeof buf, "\\u%04x", c);
                        out += buf;
                    } else {
                        out += (char)c;
                    }
            }
        }
        out += '"';
    }
    void write(std::string& out, int indent, int depth) const {
        auto newline = [&](int d) {
            if (indent >= 0) {
                out += '\n';
                out.append((size_t)(indent * d), ' ');
            }
        };
        switch (k_) {
            case kind::null: out += "null"; break;
            case kind::boolean: out += b_ ? "true" : "false"; break;
            case kind::integer: out += unsigned_ ? std::to_string(u_) : std::to_string(i_); break;
            case kind::floating: {
                if (d_ != d_ || d_ - d_ != 0.0) {  // NaN / infinity have no JSON spelling
                    out += "null";
                    break;
                }
                char buf[40];
                std::snprintf(buf, sizeof buf, "%.17g", d_);
                std::string s(buf);
                if (s.find_first_of(".eEn") == std::string::npos) s += ".0";
                out += s;
                break;
            }
            case kind::string: quote(out, s_); break;
            case kind::array:
                if (a_.empty()) {
                    out += "[]";
                    break;
                }
                out += '[';
                for (size_t i = 0; i < a_.size(); i++) {
                    if (i) out += (indent >= 0 ? "," : ", ");
                    newline(depth + 1);
                    a_[i].write(out, indent, depth + 1);
                }
                newline(depth);
                out += ']';
                break;
            case kind::object:
                if (o_.empty()) {
                    out += "{}";
                    break;
                }
                out += '{';
                for (size_t i = 0; i < o_.size(); i++) {
                    if (i) out += (indent >= 0 ? "," : ", ");
                    newline(depth + 1);
                    quote(out, o_[i].first);
                    out += ": ";
                    o_[i].second.write(out, indent, depth + 1);
                }
                newline(depth);
                out += '}';
                break;
        }
    }
};

inline std::ostream& operator<<(std::ostream& os, const json& j) { return os << j.dump(); }

}  // namespace hnh
