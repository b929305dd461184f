// pinned_vector.hpp -- a growable array of trivially copyable elements in pinned host memory (from the process-wide
// cache of alignment_impl.hpp), for the staging arrays a batch uploads: a copy from pinned memory runs at link speed
// and asynchronously, a copy from a std::vector goes through the runtime's bounce buffer.
#pragma once
#include <cstddef>
#include <cstring>
#include <type_traits>
#include <utility>

#include "alignment_impl.hpp"

namespace claraparabricks
{
namespace genomeworks
{
namespace cudaaligner
{

template <typename T>
class PinnedVector
{
    static_assert(std::is_trivially_copyable<T>::value, "PinnedVector holds plain data only");

public:
    PinnedVector() = default;
    PinnedVector(const PinnedVector&) = delete;
    PinnedVector& operator=(const PinnedVector&) = delete;
    PinnedVector(PinnedVector&& o) noexcept { swap(o); }
    PinnedVector& operator=(PinnedVector&& o) noexcept
    {
        if (this != &o)
        {
            release();
            swap(o);
        }
        return *this;
    }
    ~PinnedVector() { release(); }

    size_t size() const { return size_; }
    bool empty() const { return size_ == 0; }
    T* data() { return data_; }
    const T* data() const { return data_; }
    T& operator[](size_t i) { return data_[i]; }
    const T& operator[](size_t i) const { return data_[i]; }
    T& back() { return data_[size_ - 1]; }
    const T& back() const { return data_[size_ - 1]; }
    const T* begin() const { return data_; }
    const T* end() const { return data_ + size_; }

    void clear() { size_ = 0; }
    void reserve(size_t n)
    {
        if (n <= capacity_) return;
        size_t cap_bytes = 0;
        size_t want      = capacity_ == 0 ? n : (n > 2 * capacity_ ? n : 2 * capacity_);
        char* fresh      = pinned_acquire(want * sizeof(T), &cap_bytes);
        if (size_ != 0) std::memcpy(fresh, data_, size_ * sizeof(T));
        release_buffer();
        data_      = reinterpret_cast<T*>(fresh);
        cap_bytes_ = cap_bytes;
        capacity_  = cap_bytes / sizeof(T);
    }
    /// grows without initialising the new elements
    void resize(size_t n)
    {
        reserve(n);
        size_ = n;
    }
    void push_back(const T& v)
    {
        if (size_ == capacity_) reserve(size_ + 1);
        data_[size_++] = v;
    }
    void assign(size_t n, const T& v)
    {
        resize(n);
        for (size_t i = 0; i < n; ++i) data_[i] = v;
    }
    /// hands the buffer over (the caller returns it with pinned_release(p, *cap_bytes)); the vector is empty afterwards
    T* detach(size_t* cap_bytes)
    {
        T* p       = data_;
        *cap_bytes = cap_bytes_;
        data_      = nullptr;
        size_ = capacity_ = cap_bytes_ = 0;
        return p;
    }
    void release()
    {
        release_buffer();
        size_ = capacity_ = 0;
    }
    void swap(PinnedVector& o) noexcept
    {
        std::swap(data_, o.data_);
        std::swap(size_, o.size_);
        std::swap(capacity_, o.capacity_);
        std::swap(cap_bytes_, o.cap_bytes_);
    }

private:
    void release_buffer()
    {
        if (data_ != nullptr) pinned_release(reinterpret_cast<char*>(data_), cap_bytes_);
        data_      = nullptr;
        cap_bytes_ = 0;
    }
    T* data_          = nullptr;
    size_t size_      = 0;
    size_t capacity_  = 0;
    size_t cap_bytes_ = 0;
};

} // namespace cudaaligner
} // namespace genomeworks
} // namespace claraparabricks
