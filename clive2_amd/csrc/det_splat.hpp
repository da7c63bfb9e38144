// det_splat.hpp -- the light image of cl2_set_reproducible(1): a sort + segmented sum in a fixed order instead of float atomics.
//
// Reference: connect_paths scatters every t = 1 contribution to slot `id + s * total_pixels` of five parallel arrays
// (trace.metal:817-823), light_sort orders them by target pixel (:872-934, 300 launches of a bitonic network at 1080p),
// renderer.py:97-111 finds each pixel's run on the host and light_image_gather sums it (:937-964) -- a deterministic chain.
// The product's default replaces the chain by float atomics (same sums, order decided by the hardware: the one output of the
// pipeline that differs between two runs, by a few ulp).  With the switch on, k_connect_resolve writes the same records --
// key = target entry, value = {c.xyz, w}, at the reference's slot (s - 1) * B + source entry -- one STABLE radix sort of
// {key, slot} pairs on the key's bits alone (rocPRIM; the slots go in ascending, so a target's run comes out ordered by
// (s, source pixel)), and k_det_gather sums each target's run front to back: two renders give the same bytes.  (The order inside a run is not the bitonic network's, so the sums still differ from the reference chain's in the
// last bits; the oracle keeps the tolerance, two runs of the product do not need one.)
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>

namespace cl2 {

constexpr unsigned DET_NO_KEY = ~0u;      // slot without a contribution: sorts behind every real key

// {keys_out, slots_out} = {keys_in, slots_in} sorted ascending and stably on key bits [0, end_bit).  tmp == nullptr: only sets
// tmp_bytes.  The inputs are left as they are (slots_in is an identity table, written once).
hipError_t det_sort_pairs(void* tmp, size_t& tmp_bytes, const unsigned* keys_in, unsigned* keys_out, const unsigned* slots_in,
                          unsigned* slots_out, size_t n, unsigned end_bit, hipStream_t st);

}  // namespace cl2
