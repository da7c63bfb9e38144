// l1_gather_rate.hip -- how fast can a CU serve SCATTERED 16-byte loads that hit in its L1 / L2?
// The persistent BVH walk issues, per lane and step, two float4 loads for a node record and three per triangle, each
// lane at its own address.  This measures wave64 `global_load_dwordx4` instructions per CU-cycle as a function of
// (a) how many lanes are active and (b) how many lanes share a 128-byte line, on a table that is L2-resident
// (TABLE_KB = 4096) or L1-resident (32).  Output: cycles per load instruction per CU and lane-lines per cycle.
//   hipcc --offload-arch=gfx950 -O3 tools/l1_gather_rate.hip -o tools/l1_gather_rate && tools/l1_gather_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void gather(const float4* __restrict__ table, unsigned mask_words, int iters, int active_lanes,
                                             int lanes_per_line, float4* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    // pseudo-random walk through the table; lanes of a group of `lanes_per_line` stay in one 128-byte line (8 float4)
    unsigned s = (blockIdx.x * 256u + threadIdx.x / lanes_per_line * lanes_per_line) * 2654435761u + 12345u;
    float4 acc = make_float4(0, 0, 0, 0);
    if (lane < active_lanes) {
        for (int i = 0; i < iters; i++) {
            s = s * 1664525u + 1013904223u;
            const unsigned line = (s >> 8) & mask_words;                 // index of a 128-byte line
            const float4 v = table[(size_t)line * 8 + (lane % lanes_per_line) % 8];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    if (acc.x == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = acc;   // keep the loads alive
}

int main() {
    const int iters = 2000, blocks = 256 * 8;                            // 8 workgroups of 4 waves per CU
    for (int table_kb : {32, 4096, 262144}) {
        const size_t lines = (size_t)table_kb * 1024 / 128;
        float4 *table = nullptr, *out = nullptr;
        hipMalloc(&table, lines * 128);
        hipMalloc(&out, (size_t)blocks * 256 * sizeof(float4));
        hipMemset(table, 0, lines * 128);
        printf("table %d KB\n", table_kb);
        for (int lanes_per_line : {1, 2, 4}) {
            for (int active : {64, 32, 23, 16, 8}) {
                hipEvent_t a, b;
                hipEventCreate(&a); hipEventCreate(&b);
                hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, table, (unsigned)(lines - 1), 50, active, lanes_per_line, out);
                hipEventRecord(a);
                hipLaunchKernelGGL(gather, dim3(blocks), dim3(256), 0, 0, table, (unsigned)(lines - 1), iters, active, lanes_per_line, out);
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms = 0;
                hipEventElapsedTime(&ms, a, b);
                const double wave_loads_per_cu = (double)blocks * 4 * iters / 256.0;
                const double cycles = ms * 1e-3 * 2.4e9;
                printf("  lanes/line %d active %2d: %7.3f ms  %6.1f cycles per wave-load per CU  %5.2f lane-loads per cycle per CU  %5.2f distinct lines per cycle per CU\n",
                       lanes_per_line, active, ms, cycles / wave_loads_per_cu, active * wave_loads_per_cu / cycles,
                       (double)(active + lanes_per_line - 1) / lanes_per_line * wave_loads_per_cu / cycles);
            }
        }
        hipFree(table); hipFree(out);
    }
    return 0;
}
