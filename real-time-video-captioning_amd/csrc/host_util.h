// Host-side helpers shared by the C-ABI translation units (gitcap.hip, student.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "host_logic.h"      // pad_to, host_f2bf, e4m3 encodings, tensor table (HIP-free: also built under ASan for the CPU)

#include <cstdint>
#include <cstring>
#include <vector>

struct DevTensor {
    void* p = nullptr;
    std::vector<int64_t> shape;   // logical (unpadded) shape
    bool bf16 = false;            // a GEMM weight (stored as bf16, or as e4m3 + row scales)
    bool loaded = false;
    int64_t bytes = 0;            // device bytes held (incl. row scales)
};

// Makes a handle's device current for the duration of an entry point and restores the caller's.
struct DeviceGuard {
    int prev = -1; bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
        if (prev == dev) prev = -1;
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
