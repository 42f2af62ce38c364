// Compile-time geometries of the forward (inference) kernels.  Training kernels are specialised for SpatialNet-small only
// (configs/SpatialNet.yaml as shipped); the forward kernels are templates over this struct and are instantiated for
//   GeoS  small: dim_hidden 96,  dim_ffn 192, dim_squeeze 8,  4 heads (dh 24), conv groups 8/8   (SpatialNet.yaml:16-24)
//   GeoL  large: dim_hidden 192, dim_ffn 384, dim_squeeze 16, 4 heads (dh 48), conv groups 8/8   (the "for large" comments there)
#pragma once
#include "../../include/nbss_hip.h"

template <int H_, int FFN_, int SQ_, int HEADS_>
struct Geo {
    static constexpr int H = H_, FFN = FFN_, SQ = SQ_, HEADS = HEADS_;
    static constexpr int DH = H_ / HEADS_;   // attention head width
    static constexpr int FG = H_ / 8;        // channels per F-conv group
    static constexpr int CG = FFN_ / 8;      // channels per T-conv / GroupNorm group
    static constexpr int KS = H_ / 32;       // k-steps of an H-wide contraction
    static constexpr int MT = H_ / 16;       // 16-row tiles of an H-wide output
};
typedef Geo<96, 192, 8, 4> GeoS;
typedef Geo<192, 384, 16, 4> GeoL;

inline bool geo_is_small(const nbss_cfg& c) { return c.H == 96 && c.FFN == 192 && c.SQ == 8 && c.heads == 4; }
inline bool geo_is_large(const nbss_cfg& c) { return c.H == 192 && c.FFN == 384 && c.SQ == 16 && c.heads == 4; }
