// ntt_lds_u32.hip — instantiations of the lazy LDS NTT for 32-bit residue words.
#include "ntt14.h"

#include <atomic>
#include <cstdlib>

typedef uint32_t W;
#include "ntt_lds_dispatch.inc"

int launch_ntt_lds_u32(GpuContext *ctx, uint32_t *data, size_t vectors, uint32_t L, bool inverse) {
    return dispatch_ntt_lds(ctx, data, vectors, L, inverse);
}
