// mlp_nerf_mfma.hip -- NeRFImpl::forward (NeRF.cpp:41-126) on the gfx950 matrix cores (NRF_PREC_F16_MFMA).
#include "mlp.h"

namespace nrf {

int mlp_nerf_pack_f16(nrf_mlp *m, const std::vector<float> &hp)
{
    (void)m; (void)hp;
    return NRF_OK;
}

int mlp_nerf_forward_mfma(const nrf_mlp *m, const float *x, int xs, int64_t p, float *out, int os, hipStream_t st)
{
    (void)m; (void)x; (void)xs; (void)p; (void)out; (void)os; (void)st;
    set_error("NRF_PREC_F16_MFMA for the 8x256 NeRF MLP is not built yet; use NRF_PREC_F32");
    return NRF_ERR_UNSUPPORTED;
}

}  // namespace nrf
