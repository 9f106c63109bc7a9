// Centred RMSProp (model.py:265,355-367) as a device body, so that the update of a slice of the flat parameter buffer can run
// as its own launch (air_step_epilogue) or as extra workgroups of a backward launch that leaves most of the chip idle
// (air_lstm_step_bwd_opt / air_lstm_pointwise_bwd_opt): the step is launch bound and the update is pure HBM streaming.
#pragma once
#include "air_common.h"

__device__ __forceinline__ void rmsprop_elem(float &p, float gi_raw, float &ms, float &mg, float &mom, float lr, float decay,
                                             float momentum, float eps, float gscale) {
    // no fp contraction: the same element must come out bit-identical whichever kernel applies the update (the closing launch, a
    // rider workgroup of a BPTT launch, the epilogue of the weight-gradient tile that formed the gradient) -- and this is the
    // oracle's own rounding (torch-CPU: every op rounded separately)
#pragma clang fp contract(off)
    const float gi = gi_raw * gscale;
    const float msi = decay * ms + (1.f - decay) * gi * gi;
    const float mgi = decay * mg + (1.f - decay) * gi;
    const float mo = momentum * mom + lr * gi / sqrtf(msi - mgi * mgi + eps);
    ms = msi; mg = mgi; mom = mo;
    p -= mo;
}

// device mirror of AirRmspropSlice (include/air_hip.h); lo == hi: nothing to do
struct RmspropSlice {
    float *p; const float *g; float *ms, *mg, *mom;
    size_t lo, hi, n_model;
    const float *lr_dev;
    float lr_mult_tail, decay, momentum, eps, gscale;
};
// elements [lo, hi) of the flat buffers (lo, hi, n_model multiples of 4, buffers 16-byte aligned: checked by the host), by
// workgroup vblock of vgrid, any workgroup size
__device__ __forceinline__ void rmsprop_slice_body(const RmspropSlice &s, int vblock, int vgrid) {
    const float lr0 = s.lr_dev[0];
    float4 *p4 = reinterpret_cast<float4 *>(s.p), *ms4 = reinterpret_cast<float4 *>(s.ms), *mg4 = reinterpret_cast<float4 *>(s.mg),
           *mom4 = reinterpret_cast<float4 *>(s.mom);
    const float4 *g4 = reinterpret_cast<const float4 *>(s.g);
    const size_t q1 = s.hi >> 2, stride = (size_t)vgrid * blockDim.x;
    for (size_t q = (s.lo >> 2) + (size_t)vblock * blockDim.x + threadIdx.x; q < q1; q += stride) {
        const float lr = (q << 2) < s.n_model ? lr0 : lr0 * s.lr_mult_tail;
        float4 pv = p4[q], gv = g4[q], a = ms4[q], b = mg4[q], c = mom4[q];
        rmsprop_elem(pv.x, gv.x, a.x, b.x, c.x, lr, s.decay, s.momentum, s.eps, s.gscale);
        rmsprop_elem(pv.y, gv.y, a.y, b.y, c.y, lr, s.decay, s.momentum, s.eps, s.gscale);
        rmsprop_elem(pv.z, gv.z, a.z, b.z, c.z, lr, s.decay, s.momentum, s.eps, s.gscale);
        rmsprop_elem(pv.w, gv.w, a.w, b.w, c.w, lr, s.decay, s.momentum, s.eps, s.gscale);
        ms4[q] = a; mg4[q] = b; mom4[q] = c; p4[q] = pv;
    }
}
// host side: validate and convert; returns the number of float4 the slice holds through *nq
static inline int rmsprop_slice_from_abi(const AirRmspropSlice *o, RmspropSlice &s, size_t *nq) {
    s = RmspropSlice{};
    *nq = 0;
    if (!o || o->hi == o->lo) return AIR_OK;
    AIR_REQUIRE(o->p && o->g && o->ms && o->mg && o->mom && o->lr_dev, AIR_E_NULL);
    AIR_REQUIRE(o->lo < o->hi && o->lo % 4 == 0 && o->hi % 4 == 0 && o->n_model % 4 == 0, AIR_E_SHAPE);
    AIR_REQUIRE(air_aligned16(o->p) && air_aligned16(o->g) && air_aligned16(o->ms) && air_aligned16(o->mg) && air_aligned16(o->mom),
                AIR_E_ALIGN);
    s.p = o->p; s.g = o->g; s.ms = o->ms; s.mg = o->mg; s.mom = o->mom; s.lo = o->lo; s.hi = o->hi; s.n_model = o->n_model;
    s.lr_dev = o->lr_dev; s.lr_mult_tail = o->lr_mult_tail; s.decay = o->decay; s.momentum = o->momentum; s.eps = o->eps;
    s.gscale = o->grad_scale;
    *nq = (o->hi - o->lo) >> 2;
    return AIR_OK;
}
