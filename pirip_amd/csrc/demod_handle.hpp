// pirip_amd/csrc/demod_handle.hpp -- the demodulator handle behind include/pirip_hip.h's opaque pirip_hip_demod (library-private:
// pirip_capi.hip owns its life cycle, capture.hip runs one long capture through it).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

#include "fsk_device.hpp"
#include "fsk_plan.hpp"

struct CaptureWork;

struct pirip_hip_demod {
    pirip::FskPlan plan;
    int nstreams = 0;
    int device = 0;
    int last_hip = 0;
    // Kernel of this handle, chosen ONCE at create (the kernels keep the integrator-memory tail in different layouts,
    // so a handle never switches): 2 = wave-per-stream (fsk_demod_wave.hip), 0 = general (fsk_demod_general.hip).
    // PIRIP_KERNEL=general (or the older PIRIP_FORCE_GENERAL) forces the general kernel: the on-device cross-check.
    int kernel = 0;
    // device tables
    float *d_hann = nullptr; float2 *d_tw = nullptr; uint16_t *d_perm = nullptr; float *d_lut = nullptr;
    float2 *d_tph = nullptr; int16_t *d_teeth = nullptr; uint32_t *d_mask_dtheta = nullptr;
    float2 *d_osc_drift = nullptr; float2 *d_osc_step = nullptr; float2 *d_timing_rec = nullptr; float *d_fast_tab = nullptr;
    // device state
    float *d_Sf = nullptr; uint32_t *d_theta = nullptr; float2 *d_hist = nullptr; pirip::StreamScalars *d_scal = nullptr; float2 *d_phic = nullptr;
    // staging for the host-buffer convenience call (stream 0)
    void *d_stage_in = nullptr; size_t stage_in_bytes = 0;
    uint8_t *d_stage_bits = nullptr; float *d_stage_filt = nullptr; float *d_stage_stats = nullptr;
    int32_t *d_stage_nframes = nullptr; int64_t *d_stage_consumed = nullptr; int64_t stage_frames = 0;
    int nin0 = 0;
    // The first frame after create / reset in the oracle's own operation order (fsk_demod_general.hip: fsk_demod_exact0_kernel; shapes
    // with P == Ts): `fresh` = every stream is still in its created state; the next batch call that holds a frame runs the prologue.
    bool fresh = true;
    int exact0 = 1;                             // pirip_hip_set_exact_first_frame / PIRIP_EXACT0=0 switch it off (A/B, tests)
    int32_t *d_first = nullptr;                 // [nstreams] samples the prologue consumed in the current call
    float *d_eye = nullptr;                     // pirip_hip_enable_eye: [nstreams][8][160] |f_int| eye traces of each stream's latest frame
    struct CaptureWork *capture = nullptr;      // capture.hip: work area of pirip_hip_demod_capture, allocated on first use
};

namespace pirip {
void demod_fill_args(const pirip_hip_demod *h, DemodArgs *a);   // dims, tables, state pointers (io left to the caller)
bool demod_bind(const pirip_hip_demod *h);                      // make the handle's device current
}  // namespace pirip
