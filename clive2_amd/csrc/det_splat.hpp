// det_splat.hpp -- the light image of cl2_set_reproducible(1): a sort + segmented sum in a fixed order instead of float atomics.
//
// Reference: connect_paths scatters every t = 1 contribution to slot `id + s * total_pixels` of five parallel arrays
// (trace.metal:817-823), light_sort orders them by target pixel (:872-934, 300 launches of a bitonic network at 1080p),
// renderer.py:97-111 finds each pixel's run on the host and light_image_gather sums it (:937-964) -- a deterministic chain.
// The product's default replaces the chain by float atomics (same sums, order decided by the hardware: the one output of the
// pipeline that differs between two runs, by a few ulp).  With the switch on, k_connect_resolve writes the same records --
// key = target entry << 32 | source slot, value = {c.xyz, w} -- one radix sort (rocPRIM: stable, no atomics on data) orders
// the keys by (target, s, source pixel), and k_det_gather sums each target's run front to back: two renders give the same
// bytes.  (The order inside a run is not the bitonic network's, so the sums still differ from the reference chain's in the
// last bits; the oracle keeps the tolerance, two runs of the product do not need one.)
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace cl2 {

constexpr unsigned long long DET_NO_KEY = ~0ull;      // slot without a contribution: sorts behind every real key

// out = in sorted ascending (all 64 bits).  tmp == nullptr: only sets tmp_bytes.
hipError_t det_sort_keys(void* tmp, size_t& tmp_bytes, const unsigned long long* in, unsigned long long* out, size_t n, hipStream_t st);

}  // namespace cl2
