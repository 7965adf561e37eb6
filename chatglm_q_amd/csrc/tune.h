// Tuning knobs of the launchers.
//
// Product build (libqlinear_hip.so): every knob is the compile-time constant written at its use - the value the cited measurement
// chose - and the library reads ONE environment variable, QLINEAR_DISPATCH (dispatch_flags() below, documented in
// include/qlinear_hip.h).  Developer build (libqlinear_hip_dev.so, -DQL_DEV_TUNING): QL_TUNE(name, dflt) reads the environment
// variable `name` once, which is how the A/B sweeps under tools/ move a knob without rebuilding.
#pragma once
#include <cstdlib>

namespace ql {

#ifdef QL_DEV_TUNING
inline int tune_env(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#define QL_TUNE(name, dflt) ([]() -> int { static const int v = ::ql::tune_env(name, dflt); return v; }())
#else
#define QL_TUNE(name, dflt) (dflt)
#endif

// QLINEAR_DISPATCH: comma-separated kernel families the dispatch must NOT use (every one has a slower fallback that computes
// the same function): "no256" (256 x 256-tile GEMMs), "nopeel" (qkv_proj's second launch on the 128-row-tile kernel),
// "nofewrow" (3..32-row kernels), "norows4" (4x4x4-MFMA kernel for 2..4 rows), "nogroupattn" (grouped MFMA decode attention),
// "nohalf" (round 5: the half-tile last round of the int4g32 256-tile GEMM's launch; whole tiles only, the older peel where it applies -
// the rows both serve are bit-equal), "nof32mfma" (round 5: the fp32 matrix-instruction kernel for fp32 activations with 128+ rows; the VALU kernels instead).
enum : unsigned { QL_D_NO256 = 1, QL_D_NOPEEL = 2, QL_D_NOFEWROW = 4, QL_D_NOROWS4 = 8, QL_D_NOGROUPATTN = 16, QL_D_NOHALF = 32,
                  QL_D_NOF32MFMA = 64, QL_D_NOROWS16 = 128 };
unsigned dispatch_flags();     // abi.hip: parsed once; qlinear_dispatch_reload() parses again (tests, A/B tools)

}  // namespace ql
