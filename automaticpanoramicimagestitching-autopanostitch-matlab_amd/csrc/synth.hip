// synth.hip — bench/test support only (NOT part of the reference boundary): renders one synthetic view of
// the procedural world of synth.py (lattice value noise + sparse gaussian blobs as a function of the world
// ray) in a single kernel, so that preparing 64 4K inputs costs milliseconds instead of millions of tiny
// framework launches.  Same formulas as synth.py's torch implementation.
#include "aps_internal.h"

namespace aps {

__device__ __forceinline__ float hash3(int ix, int iy, int iz, unsigned seed) {
    unsigned h = ((unsigned)ix * 73856093u) ^ ((unsigned)iy * 19349663u) ^ ((unsigned)iz * 83492791u) ^
                 (seed * 2654435761u);
    h = (h ^ (h >> 16)) * 0x45D9F3Bu;
    h = (h ^ (h >> 16)) * 0x45D9F3Bu;
    h = h ^ (h >> 16);
    return (float)(h & 0xFFFFFFu) / 16777216.0f;
}

__device__ float value_noise(float px, float py, float pz, unsigned seed) {
    const float fx = floorf(px), fy = floorf(py), fz = floorf(pz);
    float tx = px - fx, ty = py - fy, tz = pz - fz;
    tx = tx * tx * (3 - 2 * tx);
    ty = ty * ty * (3 - 2 * ty);
    tz = tz * tz * (3 - 2 * tz);
    const int ix = (int)fx, iy = (int)fy, iz = (int)fz;
    float out = 0;
    for (int dx = 0; dx < 2; ++dx)
        for (int dy = 0; dy < 2; ++dy)
            for (int dz = 0; dz < 2; ++dz) {
                const float w = (dx ? tx : 1 - tx) * (dy ? ty : 1 - ty) * (dz ? tz : 1 - tz);
                out += w * hash3(ix + dx, iy + dy, iz + dz, seed);
            }
    return out;
}

__device__ float blobs(float px, float py, float pz, unsigned seed) {
    const int cx0 = (int)floorf(px), cy0 = (int)floorf(py), cz0 = (int)floorf(pz);
    float out = 0;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const int cx = cx0 + dx, cy = cy0 + dy, cz = cz0 + dz;
                const float ox = (float)cx + hash3(cx, cy, cz, seed + 1);
                const float oy = (float)cy + hash3(cx, cy, cz, seed + 2);
                const float oz = (float)cz + hash3(cx, cy, cz, seed + 3);
                const float amp = hash3(cx, cy, cz, seed + 4) * 2 - 1;
                const float rad = 0.12f + 0.2f * hash3(cx, cy, cz, seed + 5);
                const float d2 = (px - ox) * (px - ox) + (py - oy) * (py - oy) + (pz - oz) * (pz - oz);
                out += amp * __expf(-d2 / (2 * rad * rad));
            }
    return out;
}

struct SynthCam {
    float K[9];  // row-major
    float R[9];  // row-major world->camera
};

__global__ void synth_view_kernel(SynthCam cam, int H, int W, unsigned seed, float finest_px, float gain,
                                  uint8_t* __restrict__ out) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const float f = cam.K[0];
    const float cx = ((float)(x + 1) - cam.K[2]) / cam.K[0];
    const float cy = ((float)(y + 1) - cam.K[5]) / cam.K[4];
    // rw = R' * [cx cy 1]
    float rx = cam.R[0] * cx + cam.R[3] * cy + cam.R[6];
    float ry = cam.R[1] * cx + cam.R[4] * cy + cam.R[7];
    float rz = cam.R[2] * cx + cam.R[5] * cy + cam.R[8];
    const float inv = rsqrtf(rx * rx + ry * ry + rz * rz);
    rx *= inv;
    ry *= inv;
    rz *= inv;
    const float base = f / finest_px;
    for (int ch = 0; ch < 3; ++ch) {
        float acc = 0.5f;
        for (int o = 0; o < 7; ++o) {
            const float freq = base / (float)(1 << o);
            if (freq < 2) break;
            acc += 0.22f * (value_noise(rx * freq, ry * freq, rz * freq, seed + 17 * o + 101 * ch) - 0.5f) * (0.6f + 0.1f * o);
            if (o >= 1) {
                const float fb = freq / 3.0f;
                acc += 0.30f * 0.5f * blobs(rx * fb, ry * fb, rz * fb, seed + 31 * o + 7) * (ch ? 0.5f : 0.6f);
            }
        }
        acc *= gain;
        acc = fminf(fmaxf(acc, 0.f), 1.f);
        out[((size_t)y * W + x) * 3 + ch] = (uint8_t)(acc * 255.0f + 0.5f);
    }
}

}  // namespace aps

using namespace aps;

extern "C" int aps_synth_view(const double* K_rowmajor, const double* R_rowmajor, int height, int width,
                              unsigned seed, float finest_px, float gain, uint8_t* out) {
    return guarded([&] {
        APS_REQUIRE(K_rowmajor && R_rowmajor && out, APS_E_ARG, "NULL argument");
        APS_REQUIRE(height > 0 && width > 0 && finest_px > 0, APS_E_ARG, "bad size");
        ctx();
        SynthCam c;
        for (int e = 0; e < 9; ++e) {
            c.K[e] = (float)K_rowmajor[e];
            c.R[e] = (float)R_rowmajor[e];
        }
        Out<uint8_t> o(out, (size_t)height * width * 3);
        synth_view_kernel<<<dim3(cdiv(width, 256), height), 256, 0, stream()>>>(c, height, width, seed, finest_px, gain, o);
        check_launch("synth_view_kernel");
        o.commit();
        APS_HIP(hipStreamSynchronize(stream()));
    });
}
