// r5 probe: which instruction class of the fused head goes wrong while f16 MFMA launches run on other streams?
// Stand-alone library (hipcc --offload-arch=gfx950 -shared -fPIC), driven by tools/diag_trans_probe.py through ctypes.  Every kernel is a
// pure function of its inputs; outputs are compared bit for bit against an idle-GPU reference.
#include <hip/hip_runtime.h>

// 1: v_exp_f32 only (register data)   2: global loads only (the head's 4-tap / 48-plane pattern, L1 hits)   3: FMA only
// 4: v_rcp_f32 only                   5: v_exp_f32 on loaded data (head-like mix)                            6: polynomial exp2 (no trans unit)
__global__ __launch_bounds__(256) void probe_kernel(const float* __restrict__ in, float* __restrict__ out, int n, int plane, int nplanes, int mode) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float acc = 0.f;
    const float x0 = in[i % plane];
    if (mode == 1) {
        float t = x0 * 0.01f;
        for (int k = 0; k < 240; ++k) { acc += __builtin_amdgcn_exp2f(t - (float)k * 0.03125f); }
    } else if (mode == 2) {
        const int o = (i >> 2) % (plane - 300);
        for (int k = 0; k < nplanes; ++k) {
            const float* cp = in + (size_t)k * plane;
            acc += 0.25f * cp[o] + 0.25f * cp[o + 1] + 0.25f * cp[o + 240] + 0.25f * cp[o + 241];
        }
    } else if (mode == 3) {
        float t = x0;
        for (int k = 0; k < 480; ++k) { t = fmaf(t, 0.999f, 0.001f * (float)k); acc = fmaf(acc, 0.5f, t); }
    } else if (mode == 4) {
        float t = fabsf(x0) + 1.0f;
        for (int k = 0; k < 240; ++k) { acc += __builtin_amdgcn_rcpf(t + (float)k); }
    } else if (mode == 5) {
        const int o = (i >> 2) % (plane - 300);
        for (int k = 0; k < nplanes; ++k) {
            const float* cp = in + (size_t)k * plane;
            const float v = 0.25f * cp[o] + 0.25f * cp[o + 1] + 0.25f * cp[o + 240] + 0.25f * cp[o + 241];
            acc += __builtin_amdgcn_exp2f(v - 8.f) + __builtin_amdgcn_exp2f(v * 0.5f - 9.f) + __builtin_amdgcn_exp2f(v * 0.25f - 7.f) + __builtin_amdgcn_exp2f(-v - 9.f);
        }
    } else if (mode == 6) {
        float t = x0 * 0.01f;
        for (int k = 0; k < 240; ++k) {
            const float u = t - (float)k * 0.03125f;
            const float fl = floorf(u), f = u - fl;
            float p = 1.530277e-4f;
            p = fmaf(p, f, 1.339887e-3f); p = fmaf(p, f, 9.618437e-3f); p = fmaf(p, f, 5.550357e-2f);
            p = fmaf(p, f, 2.402265e-1f); p = fmaf(p, f, 6.931472e-1f); p = fmaf(p, f, 1.0f);
            acc += ldexpf(p, (int)fl);
        }
    }
    out[i] = acc;
}

extern "C" int probe_launch(const float* in, float* out, int n, int plane, int nplanes, int mode, void* stream) {
    hipLaunchKernelGGL(probe_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, out, n, plane, nplanes, mode);
    return (int)hipGetLastError();
}
