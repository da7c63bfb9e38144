// det_splat.hip -- the radix sort of the reproducible light image (det_splat.hpp); its own translation unit so that
// renderer_api.hip does not have to parse rocPRIM.
#include <cstring>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "det_splat.hpp"

namespace cl2 {

hipError_t det_sort_pairs(void* tmp, size_t& tmp_bytes, const unsigned* keys_in, unsigned* keys_out, const unsigned* slots_in,
                          unsigned* slots_out, size_t n, unsigned end_bit, hipStream_t st) {
    return rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_in, keys_out, slots_in, slots_out, n, 0u, end_bit, st);
}

}  // namespace cl2
