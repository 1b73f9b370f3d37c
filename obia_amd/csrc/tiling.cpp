// tiling.cpp -- B3: the tiled driver (create_tiled_segments, obia/utils/tiling.py:103-291) on label rasters.
#include "slic.hpp"

using namespace obia;

extern "C" {

int obia_tiled_slic_f32_dev(obia_ctx *ctx, const float *img, const uint8_t *mask, int H, int W, int C,
                            const obia_tiling_params *tiling, const obia_slic_params *params, int32_t *labels_out,
                            int64_t *n_segments_out) {
    (void)ctx; (void)img; (void)mask; (void)H; (void)W; (void)C; (void)tiling; (void)params; (void)labels_out; (void)n_segments_out;
    set_error("tiled driver not built yet");
    return OBIA_E_UNSUPPORTED;
}

int obia_tiled_slic_f32(obia_ctx *ctx, const float *img, const uint8_t *mask, int H, int W, int C,
                        const obia_tiling_params *tiling, const obia_slic_params *params, int32_t *labels_out,
                        int64_t *n_segments_out) {
    (void)ctx; (void)img; (void)mask; (void)H; (void)W; (void)C; (void)tiling; (void)params; (void)labels_out; (void)n_segments_out;
    set_error("tiled driver not built yet");
    return OBIA_E_UNSUPPORTED;
}

}  // extern "C"
