// tensor.hpp -- the slice of CoFHE's Tensor<T> container interface that the local
// ciphertext-tensor path and its harness use (reference: include/common/tensor.hpp:1096-1246;
// call sites benchmarks/local.cpp:83-132, include/x86_64/cpu_cryptosystem_tensor_ops.inl:199-211).
// Own implementation: a shape plus a shared, row-major std::vector.  As in the reference a
// copy is SHALLOW (copies share storage; flatten()/reshape() only change the handle's shape).
#pragma once
#include <cstddef>
#include <initializer_list>
#include <memory>
#include <stdexcept>
#include <vector>

namespace CoFHE {

template <typename T>
class Tensor {
  public:
    Tensor() : data_(std::make_shared<std::vector<T>>()), shape_{0} {}
    // zero-degree (scalar) tensor
    explicit Tensor(const T &value) : data_(std::make_shared<std::vector<T>>(1, value)), shape_{}, zero_degree_(true) {}
    Tensor(size_t n, const T &value) : data_(std::make_shared<std::vector<T>>(n, value)), shape_{n} {}
    Tensor(size_t n, size_t m, const T &value) : data_(std::make_shared<std::vector<T>>(n * m, value)), shape_{n, m} {}
    Tensor(const std::vector<size_t> &shape, const T &value) : shape_(shape) {
        data_ = std::make_shared<std::vector<T>>(count(shape), value);
    }
    Tensor(std::initializer_list<size_t> shape, const T &value) : Tensor(std::vector<size_t>(shape), value) {}

    size_t ndim() const { return zero_degree_ ? 0 : shape_.size(); }
    const std::vector<size_t> &shape() const { return shape_; }
    size_t num_elements() const { return data_->size(); }
    size_t size() const { return data_->size(); }
    bool is_zero_degree() const { return zero_degree_; }
    bool is_column_vector() const { return !zero_degree_ && shape_.size() == 1; }
    T get_value() const {
        if (!zero_degree_) throw std::runtime_error("Tensor is not zero degree");
        return (*data_)[0];
    }

    void flatten() {
        if (zero_degree_) throw std::runtime_error("Cannot flatten a zero degree tensor");
        shape_ = {data_->size()};
    }
    void reshape(const std::vector<size_t> &shape) {
        if (zero_degree_) throw std::runtime_error("Cannot reshape a zero degree tensor");
        if (count(shape) != data_->size()) throw std::runtime_error("Invalid shape");
        shape_ = shape;
    }
    void reshape(std::initializer_list<size_t> shape) { reshape(std::vector<size_t>(shape)); }

    // element access by row-major flat index (the only form the hot path and harness use)
    T &at(size_t i) {
        if (i >= data_->size()) throw std::out_of_range("Index out of range");
        return (*data_)[i];
    }
    const T &at(size_t i) const {
        if (i >= data_->size()) throw std::out_of_range("Index out of range");
        return (*data_)[i];
    }
    // 2-D access (the Beaver-triplet tensors are read as triplets.at(i, j),
    // include/smpc/ciphertext_multiplications.hpp:123-125)
    T &at(size_t i, size_t j) {
        if (shape_.size() != 2 || i >= shape_[0] || j >= shape_[1]) throw std::out_of_range("Index out of range");
        return (*data_)[i * shape_[1] + j];
    }
    const T &at(size_t i, size_t j) const {
        if (shape_.size() != 2 || i >= shape_[0] || j >= shape_[1]) throw std::out_of_range("Index out of range");
        return (*data_)[i * shape_[1] + j];
    }
    T &operator[](size_t i) { return (*data_)[i]; }
    const T &operator[](size_t i) const { return (*data_)[i]; }

  private:
    static size_t count(const std::vector<size_t> &s) {
        size_t n = 1;
        for (size_t d : s) n *= d;
        return n;
    }
    std::shared_ptr<std::vector<T>> data_;
    std::vector<size_t> shape_;
    bool zero_degree_ = false;
};

}  // namespace CoFHE
