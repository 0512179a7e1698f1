// GPU voxeliser: atoms of a structure + one local frame per residue -> per-residue voxel
// frames [n_res, V, V, V, C], the input of the CNN (what aposteriori.make_frame_dataset produces for the reference:
// ui.py:73-86, README.md:83-97; dataset layout design_utils/utils.py:238-251).
//
// PARITY UNPINNED against aposteriori==2.4.0 (source not in the reference tree): the kernel implements the
// specification written out in timed_hip/voxeliser.py (items 3-6) and is tested against its NumPy restatement
// (oracle/voxel_oracle.py) — bit-exact for boolean frames, to exp() rounding for Gaussian ones.
//
// One workgroup per residue frame.  Phase 1 walks the structure's atoms in order, transforms each into the residue's
// frame (float32, separate multiplies and adds: no FMA contraction, so the host restatement reproduces the indices bit
// for bit) and keeps those whose voxel lies inside the cube in an LDS list — an ordered compaction (wave ballots + a
// per-chunk wave prefix), so the list is in atom order.  Phase 2 is a GATHER: every thread owns voxels and sums the
// contributions of the listed atoms in list order — deterministic, no atomics (a scatter with float atomics would make
// overlapping Gaussians of neighbouring backbone atoms order-dependent).  A 21 A cube holds 150-250 encodable atoms.
#include "common.h"

namespace {

constexpr int kMaxList = 2048;

struct VoxArgs {
    const float* xyz; const int* chn; const float* sig; long long n_atoms;
    const float* frt; int V; float a; int C; int gaussian;
    void* out; int* overflow;
};

__global__ void __launch_bounds__(256) k_voxelise(const VoxArgs p) {
#pragma clang fp contract(off)
    __shared__ float Lx[kMaxList], Ly[kMaxList], Lz[kMaxList], Lk[kMaxList], Lt[kMaxList];   // local xyz, 1/(2 sigma^2), sum of the 27 weights
    __shared__ int Li[kMaxList];          // (i0 << 20) | (i1 << 10) | i2, channel in the top bits
    __shared__ int wave_cnt[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x, V = p.V, centre = V / 2;
    const float* F = p.frt + (size_t)r * 12;
    const float R00 = F[0], R01 = F[1], R02 = F[2], R10 = F[3], R11 = F[4], R12 = F[5], R20 = F[6], R21 = F[7], R22 = F[8];
    const float c0 = F[9], c1 = F[10], c2 = F[11];
    int count = 0;
    for (long long base = 0; base < p.n_atoms; base += 256) {
        const long long ai = base + tid;
        bool ok = false;
        float l0 = 0.f, l1 = 0.f, l2 = 0.f;
        int i0 = 0, i1 = 0, i2 = 0, ch = 0;
        if (ai < p.n_atoms) {
            const float d0 = p.xyz[3 * ai] - c0, d1 = p.xyz[3 * ai + 1] - c1, d2 = p.xyz[3 * ai + 2] - c2;
            l0 = (R00 * d0 + R01 * d1) + R02 * d2;
            l1 = (R10 * d0 + R11 * d1) + R12 * d2;
            l2 = (R20 * d0 + R21 * d1) + R22 * d2;
            i0 = (int)floorf(l0 / p.a + 0.5f) + centre;
            i1 = (int)floorf(l1 / p.a + 0.5f) + centre;
            i2 = (int)floorf(l2 / p.a + 0.5f) + centre;
            ch = p.chn[ai];
            ok = i0 >= 0 && i0 < V && i1 >= 0 && i1 < V && i2 >= 0 && i2 < V && ch >= 0 && ch < p.C;
        }
        const unsigned long long mask = __ballot(ok);
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(mask);
        __syncthreads();
        int off = count + before;
        for (int w = 0; w < wave; ++w) off += wave_cnt[w];
        const int chunk_total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        if (ok && off < kMaxList) {
            Lx[off] = l0; Ly[off] = l1; Lz[off] = l2;
            Li[off] = (ch << 27) | (i0 << 18) | (i1 << 9) | i2;
            const float s = p.gaussian ? p.sig[ai] : 1.f;
            Lk[off] = 1.0f / (2.0f * s * s);
        }
        count += chunk_total;
        __syncthreads();
    }
    if (count > kMaxList) { if (tid == 0) atomicExch(p.overflow, count); count = kMaxList; }
    // the 27 weights of an atom sum to Lt (added in the fixed order axis0, axis1, axis2 outer to inner)
    if (p.gaussian) {
        for (int e = tid; e < count; e += 256) {
            const int pk = Li[e];
            const int i0 = (pk >> 18) & 511, i1 = (pk >> 9) & 511, i2 = pk & 511;
            float total = 0.f;
            for (int a0 = -1; a0 <= 1; ++a0)
                for (int a1 = -1; a1 <= 1; ++a1)
                    for (int a2 = -1; a2 <= 1; ++a2) {
                        const float e0 = (float)(i0 + a0 - centre) * p.a - Lx[e];
                        const float e1 = (float)(i1 + a1 - centre) * p.a - Ly[e];
                        const float e2 = (float)(i2 + a2 - centre) * p.a - Lz[e];
                        const float r2 = (e0 * e0 + e1 * e1) + e2 * e2;
                        total = total + expf(-(r2 * Lk[e]));
                    }
            Lt[e] = total;
        }
    }
    __syncthreads();
    const int V3 = V * V * V;
    for (int v = tid; v < V3; v += 256) {
        const int x0 = v / (V * V), rem = v - x0 * V * V, x1 = rem / V, x2 = rem - x1 * V;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int e = 0; e < count; ++e) {
            const int pk = Li[e];
            const int i0 = (pk >> 18) & 511, i1 = (pk >> 9) & 511, i2 = pk & 511, ch = (pk >> 27) & 15;
            const int a0 = x0 - i0, a1 = x1 - i1, a2 = x2 - i2;
            if (p.gaussian) {
                if (a0 < -1 || a0 > 1 || a1 < -1 || a1 > 1 || a2 < -1 || a2 > 1) continue;
                const float e0 = (float)(x0 - centre) * p.a - Lx[e];
                const float e1 = (float)(x1 - centre) * p.a - Ly[e];
                const float e2 = (float)(x2 - centre) * p.a - Lz[e];
                const float r2 = (e0 * e0 + e1 * e1) + e2 * e2;
                const float w = expf(-(r2 * Lk[e])) / Lt[e];
#pragma unroll
                for (int c = 0; c < 8; ++c) if (c == ch) acc[c] = acc[c] + w;
            } else if ((a0 | a1 | a2) == 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) if (c == ch) acc[c] = 1.f;
            }
        }
        const size_t o = ((size_t)r * V3 + v) * p.C;
        if (p.gaussian) { for (int c = 0; c < p.C; ++c) ((float*)p.out)[o + c] = acc[c < 8 ? c : 0]; }
        else { for (int c = 0; c < p.C; ++c) ((unsigned char*)p.out)[o + c] = acc[c < 8 ? c : 0] != 0.f ? 1 : 0; }
    }
}

}  // namespace

extern "C" int th_voxelise(int device, const float* atoms_xyz, const int32_t* atom_channel, const float* atom_sigma, int64_t n_atoms,
                           const float* frames_rt, int64_t n_res, int voxels_per_side, float frame_edge_length, int n_channels,
                           int gaussian, void* out, int out_on_device) {
    if (n_res < 0 || n_atoms < 0 || (n_res && (!frames_rt || !out)) || (n_atoms && (!atoms_xyz || !atom_channel)))
        TH_FAIL(TH_EINVAL, "th_voxelise: null argument");
    if (gaussian && n_atoms && !atom_sigma) TH_FAIL(TH_EINVAL, "th_voxelise: Gaussian frames need per-atom sigmas");
    if (voxels_per_side < 1 || voxels_per_side > 511 || !(voxels_per_side & 1)) TH_FAIL(TH_EINVAL, "th_voxelise: voxels_per_side must be odd and < 512");
    if (n_channels < 1 || n_channels > 8) TH_FAIL(TH_EUNSUP, "th_voxelise: 1..8 channels");
    if (!(frame_edge_length > 0.f)) TH_FAIL(TH_EINVAL, "th_voxelise: frame_edge_length");
    if (n_res == 0) return TH_OK;
    HIP_TRY(hipSetDevice(device));
    const size_t V3 = (size_t)voxels_per_side * voxels_per_side * voxels_per_side;
    const size_t out_bytes = (size_t)n_res * V3 * n_channels * (gaussian ? sizeof(float) : 1);
    float *d_xyz = nullptr, *d_sig = nullptr, *d_frt = nullptr;
    int *d_chn = nullptr, *d_flag = nullptr;
    void* d_out = out_on_device ? out : nullptr;
    int rc = TH_OK;
    auto fail = [&](hipError_t e, const char* what) { th_set_error("th_voxelise: %s: %s", what, hipGetErrorString(e)); rc = TH_EHIP; };
    hipError_t e;
    do {
        const size_t na = (size_t)std::max<int64_t>(n_atoms, 1);
        if ((e = th_malloc_retry(&d_xyz, na * 12)) != hipSuccess || (e = th_malloc_retry(&d_chn, na * 4)) != hipSuccess ||
            (e = th_malloc_retry(&d_sig, na * 4)) != hipSuccess || (e = th_malloc_retry(&d_frt, (size_t)n_res * 48)) != hipSuccess ||
            (e = th_malloc_retry(&d_flag, 4)) != hipSuccess) { fail(e, "hipMalloc"); break; }
        if (!out_on_device && (e = th_malloc_retry(&d_out, out_bytes)) != hipSuccess) { fail(e, "th_malloc_retry(frames)"); break; }
        if (n_atoms) {
            if ((e = hipMemcpy(d_xyz, atoms_xyz, (size_t)n_atoms * 12, hipMemcpyHostToDevice)) != hipSuccess ||
                (e = hipMemcpy(d_chn, atom_channel, (size_t)n_atoms * 4, hipMemcpyHostToDevice)) != hipSuccess) { fail(e, "upload"); break; }
            if (atom_sigma && (e = hipMemcpy(d_sig, atom_sigma, (size_t)n_atoms * 4, hipMemcpyHostToDevice)) != hipSuccess) { fail(e, "upload"); break; }
        }
        if ((e = hipMemcpy(d_frt, frames_rt, (size_t)n_res * 48, hipMemcpyHostToDevice)) != hipSuccess ||
            (e = hipMemset(d_flag, 0, 4)) != hipSuccess) { fail(e, "upload"); break; }
        VoxArgs a{d_xyz, d_chn, d_sig, (long long)n_atoms, d_frt, voxels_per_side, frame_edge_length / (float)voxels_per_side,
                  n_channels, gaussian, d_out, d_flag};
        hipLaunchKernelGGL(k_voxelise, dim3((unsigned)n_res), dim3(256), 0, 0, a);
        if ((e = hipGetLastError()) != hipSuccess || (e = hipDeviceSynchronize()) != hipSuccess) { fail(e, "kernel"); break; }
        int flag = 0;
        if ((e = hipMemcpy(&flag, d_flag, 4, hipMemcpyDeviceToHost)) != hipSuccess) { fail(e, "download"); break; }
        if (flag) { th_set_error("th_voxelise: %d encodable atoms inside one frame (limit %d)", flag, kMaxList); rc = TH_EUNSUP; break; }
        if (!out_on_device && (e = hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost)) != hipSuccess) { fail(e, "download"); break; }
    } while (false);
    (void)hipFree(d_xyz); (void)hipFree(d_chn); (void)hipFree(d_sig); (void)hipFree(d_frt); (void)hipFree(d_flag);
    if (!out_on_device) (void)hipFree(d_out);
    return rc;
}
