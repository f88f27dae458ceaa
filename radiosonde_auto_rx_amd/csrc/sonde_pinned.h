// sonde_pinned.h — a host buffer the device copies into asynchronously: page-locked memory of its own (hipHostMalloc).
// (Registering a std::vector's storage — hipHostRegister — pins whole PAGES: two small buffers of two objects can share one, and the first to be
//  unregistered unpins it under the other.  The engines' host-side landing buffers are therefore allocated, not registered.)
#ifndef SONDE_PINNED_H
#define SONDE_PINNED_H
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <string.h>

template <class T> struct Pinned {
    T *p = nullptr; size_t n = 0;
    Pinned() = default;
    Pinned(const Pinned &) = delete;
    Pinned &operator=(const Pinned &) = delete;
    ~Pinned() { release(); }
    // -> false when the allocation fails (nothing is held then)
    bool alloc(size_t count) {
        release();
        if (hipHostMalloc((void **)&p, (count ? count : 1) * sizeof(T), hipHostMallocDefault) != hipSuccess) { p = nullptr; (void)hipGetLastError(); return false; }
        memset((void *)p, 0, (count ? count : 1) * sizeof(T));
        n = count;
        return true;
    }
    void release() { if (p) (void)hipHostFree(p); p = nullptr; n = 0; }
    T *data() { return p; }
    const T *data() const { return p; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T &operator[](size_t i) { return p[i]; }
    const T &operator[](size_t i) const { return p[i]; }
    T *begin() { return p; }
    T *end() { return p + n; }
};
#endif
