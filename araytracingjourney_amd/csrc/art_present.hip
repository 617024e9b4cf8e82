// art_present.hip -- the step right after the path: what the reference stores and shows.
//   pack:    colour / normal -> B10G11R11_UFLOAT_PACK32 (renderer.rs:268, vk_rt_lightning_shadows.rs:152), depth -> R16_SFLOAT (:142)
//   present: VkTonemap::present (vk_tonemap.rs:469-552) = shaders/tonemap/tonemap.comp.glsl:29-40:
//            colour (read back from the packed image) * ao/255 -> LpmFilter(LPM_CONFIG_709_709) -> pow(1/2.2) -> B8G8R8A8_UNORM
// LpmSetup is the host-side port of vk_tonemap.rs:122-325 (itself a port of ffx_lpm.h's LpmSetup); LpmMap follows
// ffx_lpm.h:727-832 with every path flag false (LPM_CONFIG_709_709, ffx_lpm.h:616).
#include "art_internal.h"
#include <cmath>

namespace art {

__device__ inline uint16_t pack_f16(float f) {
    uint32_t u = __float_as_uint(f);
    uint32_t sign = (u >> 16) & 0x8000u, e8 = (u >> 23) & 255u, m = u & 0x7FFFFFu;
    if (e8 == 255u) return (uint16_t)(sign | 0x7C00u | (m ? 0x200u : 0u));
    int e = (int)e8 - 127 + 15;
    if (e >= 31) return (uint16_t)(sign | 0x7C00u);
    int shift = 13; uint32_t full = m | (e8 ? 0x800000u : 0u);
    if (e <= 0) { shift += 1 - e; e = 0; if (shift > 31) return (uint16_t)sign; } else full &= 0x7FFFFFu;
    uint32_t q = full >> shift, rem = full & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (((uint32_t)e << 10) + q));
}

struct LpmCtl { float sat[3], contrast, tsb[2], lumaT[3], crosstalk[3], rcpLumaT[3]; };

__device__ inline float satf(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

__global__ __launch_bounds__(256) void k_present(uint32_t n, const float4 *__restrict__ color, const float4 *__restrict__ normal, const float *__restrict__ depth,
                                                 const uint32_t *__restrict__ ao, LpmCtl L, uint32_t *__restrict__ pcolor, uint32_t *__restrict__ pnormal,
                                                 uint16_t *__restrict__ pdepth, uint32_t *__restrict__ bgra) {
    uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    float4 c4 = color[i], n4 = normal[i];
    uint32_t pk = pack_b10g11r11(c4.x, c4.y, c4.z);
    pcolor[i] = pk;
    pnormal[i] = pack_b10g11r11(n4.x, n4.y, n4.z);
    pdepth[i] = pack_f16(depth[i]);
    float R = unpack_ufloat<6>(pk & 0x7FFu), G = unpack_ufloat<6>((pk >> 11) & 0x7FFu), B = unpack_ufloat<5>(pk >> 22); // tonemap.comp.glsl:32
    float a = (float)(ao ? ao[i] : 255u) / 255.0f;                                                                      // :33-34
    R *= a; G *= a; B *= a;
    if (fmaxf(fmaxf(R, G), B) > 0.0f) { // LpmMap, ffx_lpm.h:727-832 (no shoulder, con, soft, con2, clip, scaleOnly)
        float rcpMax = 1.0f / fmaxf(fmaxf(R, G), B);
        float ratioR = powf(R * rcpMax, L.sat[0]), ratioG = powf(G * rcpMax, L.sat[1]), ratioB = powf(B * rcpMax, L.sat[2]);
        float luma = G * L.lumaT[1] + (R * L.lumaT[0] + (B * L.lumaT[2]));
        luma = powf(luma, L.contrast);
        luma = luma * (1.0f / (luma * L.tsb[0] + L.tsb[1]));
        float lumaRatio = ratioR * L.lumaT[0] + ratioG * L.lumaT[1] + ratioB * L.lumaT[2];
        float ratioScale = satf(luma * (1.0f / lumaRatio));
        R = satf(ratioR * ratioScale); G = satf(ratioG * ratioScale); B = satf(ratioB * ratioScale);
        float capR = -L.crosstalk[0] * R + L.crosstalk[0], capG = -L.crosstalk[1] * G + L.crosstalk[1], capB = -L.crosstalk[2] * B + L.crosstalk[2];
        float lumaAdd = satf((-B) * L.lumaT[2] + ((-R) * L.lumaT[0] + ((-G) * L.lumaT[1] + luma)));
        float t = lumaAdd * (1.0f / (capG * L.lumaT[1] + (capR * L.lumaT[0] + (capB * L.lumaT[2]))));
        R = satf(t * capR + R); G = satf(t * capG + G); B = satf(t * capB + B);
        lumaAdd = satf((-B) * L.lumaT[2] + ((-R) * L.lumaT[0] + ((-G) * L.lumaT[1] + luma)));
        R = satf(lumaAdd * L.rcpLumaT[0] + R); G = satf(lumaAdd * L.rcpLumaT[1] + G); B = satf(lumaAdd * L.rcpLumaT[2] + B);
    } else { R = 0.f; G = 0.f; B = 0.f; }
    R = powf(R, 1.0f / 2.2f); G = powf(G, 1.0f / 2.2f); B = powf(B, 1.0f / 2.2f);                                      // rgb_to_srgb_approx, color_spaces.glsl:68-70
    uint32_t r8 = (uint32_t)(satf(R) * 255.0f + 0.5f), g8 = (uint32_t)(satf(G) * 255.0f + 0.5f), b8 = (uint32_t)(satf(B) * 255.0f + 0.5f);
    bgra[i] = b8 | (g8 << 8) | (r8 << 16) | (255u << 24);                                                                // B8G8R8A8_UNORM swapchain (renderer.rs:191-199)
}

// ---- host: LpmData::new / get_control_block (vk_tonemap.rs:54-325) for LPM_CONFIG_709_709 + LPM_COLORS_709_709 -------------
static void mat3_inverse(const float m[9], float o[9]) {
    float a = m[0], b = m[1], c = m[2], d = m[3], e = m[4], f = m[5], g = m[6], h = m[7], i = m[8];
    float A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g, det = a * A + b * B + c * C, id = 1.0f / det;
    o[0] = A * id; o[1] = -(b * i - c * h) * id; o[2] = (b * f - c * e) * id;
    o[3] = B * id; o[4] = (a * i - c * g) * id; o[5] = -(a * f - c * d) * id;
    o[6] = C * id; o[7] = -(a * h - b * g) * id; o[8] = (a * e - b * d) * id;
}
static void col_rgb_to_xyz(const float r[2], const float g[2], const float b[2], const float w[2], float out[9]) { // vk_tonemap.rs:12-47
    auto xy_to_z = [](const float s[2], float o[3]) { o[0] = s[0]; o[1] = s[1]; o[2] = 1.0f - s[0] + s[1]; };       // as written in the reference (:12-14)
    float rz[3], gz[3], bz[3], w3[3];
    xy_to_z(r, rz); xy_to_z(g, gz); xy_to_z(b, bz); xy_to_z(w, w3);
    float rgb3[9] = {rz[0], gz[0], bz[0], rz[1], gz[1], bz[1], rz[2], gz[2], bz[2]};
    float rw = 1.0f / w[1];
    for (int k = 0; k < 3; k++) w3[k] *= rw;
    float inv[9]; mat3_inverse(rgb3, inv);
    float s[3];
    for (int k = 0; k < 3; k++) s[k] = inv[3 * k] * w3[0] + inv[3 * k + 1] * w3[1] + inv[3 * k + 2] * w3[2];
    for (int row = 0; row < 3; row++) for (int k = 0; k < 3; k++) out[3 * row + k] = rgb3[3 * row + k] * s[k];
}
void lpm_control_block(bool shoulder, float soft_gap, float hdr_max, float exposure, float contrast, float shoulder_contrast, const float saturation[3],
                       const float crosstalk[3], uint32_t ctl[96]) {
    (void)shoulder; (void)soft_gap; // LPM_CONFIG_709_709: no soft gamut mapping, no conversion matrices
    std::memset(ctl, 0, 96 * 4);
    contrast += 1.0f;
    float sat[3] = {saturation[0] + contrast, saturation[1] + contrast, saturation[2] + contrast};
    float mid_in = hdr_max * 0.18f * std::exp2(-exposure), mid_out = 0.18f, cs = contrast * shoulder_contrast;
    float z0 = -std::pow(mid_in, contrast), z1 = std::pow(hdr_max, cs) * std::pow(mid_in, contrast), z2 = std::pow(hdr_max, contrast) * std::pow(mid_in, cs) * mid_out;
    float z3 = std::pow(hdr_max, cs) * mid_out, z4 = std::pow(mid_in, cs) * mid_out;
    float f[40]; std::memset(f, 0, sizeof f);
    f[4] = -((z0 + (mid_out * (z1 - z2)) * (1.0f / (z3 - z4))) * (1.0f / z4));
    f[5] = (z1 - z2) * (1.0f / (z3 - z4));
    const float R[2] = {0.64f, 0.33f}, G[2] = {0.30f, 0.60f}, B[2] = {0.15f, 0.06f}, W[2] = {0.3127f, 0.3290f};
    float m[9]; col_rgb_to_xyz(R, G, B, W, m);
    float rs = 1.0f / (m[3] + m[4] + m[5]);
    float lumaT[3] = {m[3], m[4], m[5]};
    float rt = 1.0f / (lumaT[0] + lumaT[1] + lumaT[2]);
    for (int k = 0; k < 3; k++) lumaT[k] *= rt;
    f[0] = sat[0]; f[1] = sat[1]; f[2] = sat[2]; f[3] = contrast;
    f[6] = lumaT[0]; f[7] = lumaT[1]; f[8] = lumaT[2]; f[9] = crosstalk[0]; f[10] = crosstalk[1]; f[11] = crosstalk[2];
    f[12] = 1.0f / lumaT[0]; f[13] = 1.0f / lumaT[1]; f[14] = 1.0f / lumaT[2];
    f[24] = shoulder_contrast; f[25] = m[3] * rs; f[26] = m[4] * rs; f[27] = m[5] * rs;
    std::memcpy(ctl, f, 40 * 4); // ctl[0..9]; the packed fp16 half (ctl[16..20]) serves LpmFilterH only and is left zero
}

void launch_present(uint32_t n, const float4 *color, const float4 *normal, const float *depth, const uint32_t *ao, const uint32_t ctl[96], uint32_t *pcolor,
                    uint32_t *pnormal, uint16_t *pdepth, uint32_t *bgra, hipStream_t s) {
    float f[40]; std::memcpy(f, ctl, 40 * 4);
    LpmCtl L;
    for (int k = 0; k < 3; k++) { L.sat[k] = f[k]; L.lumaT[k] = f[6 + k]; L.crosstalk[k] = f[9 + k]; L.rcpLumaT[k] = f[12 + k]; }
    L.contrast = f[3]; L.tsb[0] = f[4]; L.tsb[1] = f[5];
    if (n) k_present<<<(n + 255) / 256, 256, 0, s>>>(n, color, normal, depth, ao, L, pcolor, pnormal, pdepth, bgra);
}

} // namespace art
