// record_chase.hip -- what does a dependent fetch of a tree-node record cost as a function of its SIZE, when every lane
// chases its own pointers (the persistent BVH walks)?  Each lane follows a pseudo-random chain through a table of
// records; a step loads K consecutive float4 (K x 16 bytes, record aligned to its size) and derives the next index from
// what it read, so a step cannot start before the previous one has returned.  8 workgroups of 4 waves per CU (the
// occupancy of the walks).  Tables: 2 MB (L2-resident), 128 MB (Infinity Cache), 2 GB (HBM).
// Output per (table, K): ns per step and lane-chain, steps per second for the whole chip, bytes per second moved.
//   hipcc --offload-arch=gfx950 -O3 tools/record_chase.hip -o tools/record_chase && tools/record_chase
#include <hip/hip_runtime.h>
#include <cstdio>

template <int K>
__global__ __launch_bounds__(256) void chase(const float4* __restrict__ table, unsigned n_records, int steps, unsigned* __restrict__ out) {
    unsigned idx = (blockIdx.x * 256u + threadIdx.x) * 2654435761u % n_records;
    unsigned acc = 0;
    for (int i = 0; i < steps; i++) {
        const float4* rec = table + (size_t)idx * K;
        unsigned h = 0;
#pragma unroll
        for (int k = 0; k < K; k++) {
            const float4 v = rec[k];
            h += __float_as_uint(v.x) + __float_as_uint(v.w);
        }
        acc += h;
        idx = (idx * 1664525u + 1013904223u + h) % n_records;       // h is 0 (zeroed table) but the compiler cannot know
    }
    if (acc == 0x12345u) out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int K>
void run(const float4* table, size_t table_bytes, unsigned* out, const char* name) {
    const unsigned n_records = (unsigned)(table_bytes / (16 * K));
    const int blocks = 256 * 8, steps = 400;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(chase<K>, dim3(blocks), dim3(256), 0, 0, table, n_records, 20, out);
    hipEventRecord(a);
    hipLaunchKernelGGL(chase<K>, dim3(blocks), dim3(256), 0, 0, table, n_records, steps, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    const double lanes = (double)blocks * 256, total = lanes * steps;
    printf("  %-8s record %3d B: %8.3f ms  %7.1f ns per step per lane  %6.2f G steps/s  %6.2f TB/s of records\n", name, 16 * K, ms,
           ms * 1e6 / steps, total / (ms * 1e-3) / 1e9, total * 16 * K / (ms * 1e-3) / 1e12);
    hipEventDestroy(a); hipEventDestroy(b);
}

int main() {
    unsigned* out = nullptr;
    hipMalloc(&out, (size_t)256 * 8 * 256 * 4);
    const size_t sizes[] = {(size_t)2 << 20, (size_t)24 << 20, (size_t)128 << 20, (size_t)2048 << 20};
    const char* names[] = {"2 MB", "24 MB", "128 MB", "2 GB"};
    for (int t = 0; t < 4; t++) {
        float4* table = nullptr;
        if (hipMalloc(&table, sizes[t]) != hipSuccess) { printf("alloc failed\n"); return 1; }
        hipMemset(table, 0, sizes[t]);
        hipDeviceSynchronize();
        printf("table %s\n", names[t]);
        run<1>(table, sizes[t], out, names[t]);
        run<2>(table, sizes[t], out, names[t]);
        run<4>(table, sizes[t], out, names[t]);
        run<7>(table, sizes[t], out, names[t]);
        run<8>(table, sizes[t], out, names[t]);
        hipFree(table);
    }
    return 0;
}
