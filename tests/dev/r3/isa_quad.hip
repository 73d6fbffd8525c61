#include <hip/hip_runtime.h>
#include "dp_wrap.hip.inc"
#include "dp_quad.hip.inc"
template <int C> __global__ __launch_bounds__(64, 4) void kq(const DpQuad *qq, uint8_t *cells, int maxrows, int *out)
{
    DpQuad q = qq[blockIdx.x];
    for (int g = 0; g < 4; g++) { q.base[g] = uni(q.base[g]); q.rows[g] = uni(q.rows[g]); q.U[g] = uni(q.U[g]); }
    q.n = uni(q.n);
    int best[4][2][3];
    dp_forward2p_g16<C>(q, 1, 1, 3, 1, 3, 1, cells, uni(maxrows), best);
    if (lane_id() == 0) out[blockIdx.x] = best[0][0][0] + best[1][1][1] + best[2][0][2] + best[3][1][0];
}
template __global__ void kq<2>(const DpQuad *, uint8_t *, int, int *);
template __global__ void kq<4>(const DpQuad *, uint8_t *, int, int *);
template __global__ void kq<7>(const DpQuad *, uint8_t *, int, int *);
template __global__ void kq<8>(const DpQuad *, uint8_t *, int, int *);
