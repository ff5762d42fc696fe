// DArray.h — RAII device buffer of the drop-in API (reference: src/DArray.h:21-54).
//
// Same contract as the reference type: element types int / float / float3 only, zero-filled on
// construction, non-copyable, addr(offset) hands out a raw DEVICE pointer, storage released when
// the last owner goes away.  Backed by hipMalloc; fills and copies run on sphx::stream().
// Extension: swap() exchanges storage with a same-length array (used for ping-pong buffers).
// Unlike every other HIP error (printed, execution continues — global.h), a failed allocation throws
// sphx::DeviceAllocError.
#pragma once

#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include "global.h"

namespace sphx {
// thrown when device memory cannot be obtained: continuing with a null buffer would turn an
// out-of-memory condition into a GPU fault (the C ABI maps it to SPHX_ERR_HIP)
struct DeviceAllocError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
}  // namespace sphx

template <typename T>
class DArray {
    static_assert(std::is_same<T, float3>::value || std::is_same<T, float>::value ||
                      std::is_same<T, int>::value,
                  "DArray holds int, float or float3 elements only.");

public:
    explicit DArray(const unsigned int length) : _length(length), _store(allocate(length)) { clear(); }

    DArray(const DArray&) = delete;
    DArray& operator=(const DArray&) = delete;

    T* addr(const int offset = 0) const { return _store.get() + offset; }
    unsigned int length() const { return _length; }

    void clear()
    {
        if (_length) HIP_CALL(hipMemsetAsync(_store.get(), 0, sizeof(T) * _length, sphx::stream()));
    }

    void swap(DArray& other)
    {
        std::swap(_length, other._length);
        _store.swap(other._store);
    }

    ~DArray() noexcept {}

private:
    static std::shared_ptr<T> allocate(unsigned int length)
    {
        void* raw = nullptr;
        const size_t bytes = sizeof(T) * (size_t)(length ? length : 1u);
        const hipError_t e = hipMalloc(&raw, bytes);
        if (e != hipSuccess || !raw) {
            ::sphx::report_hip_error(e, __FILE__, __LINE__);
            throw ::sphx::DeviceAllocError("DArray: hipMalloc of " + std::to_string(bytes) + " bytes failed");
        }
        return std::shared_ptr<T>(static_cast<T*>(raw), [](T* p) { HIP_CALL(hipFree(p)); });
    }

    unsigned int _length;
    std::shared_ptr<T> _store;
};
