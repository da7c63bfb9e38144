// det_splat.hip -- the radix sort of the reproducible light image (det_splat.hpp); its own translation unit so that
// renderer_api.hip does not have to parse rocPRIM.
#include <cstring>
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "det_splat.hpp"

namespace cl2 {

hipError_t det_sort_keys(void* tmp, size_t& tmp_bytes, const unsigned long long* in, unsigned long long* out, size_t n, hipStream_t st) {
    return rocprim::radix_sort_keys(tmp, tmp_bytes, in, out, n, 0, 64, st);
}

}  // namespace cl2
