// knobs.hpp -- environment switches.
//
// The shipped library reads a handful of documented switches, each parsed once (product_switch):
//   MFX_HOST_PLAN=1     build the stripe/task layout on the host instead of the device (prep.hip)
//   MFX_HOST_INIT=1     draw the initial factors on the host
//   MFX_HOST_THREADS=n  worker threads of the host builder
//   MFX_PLAN_TIMING=1   print the phases of the pre-processing to stderr
//   MFX_PREDICT_CACHE=1 keep the model array of utility_predict resident in HBM between calls (opt-in)
//   MFX_DEVICES=g       mf::utility_train shards the job over g GPUs of the node (csrc/job.cpp)
// Everything else that rounds 1-2 tuned through the environment (launch width, task sizes, fold constants ...) is an
// EXPERIMENT knob: it exists only in `make variant VFLAGS=-DMFX_EXPERIMENTS` builds and is a constant in the product.
#pragma once
#include <cstdlib>

namespace mfx {

inline int env_int_raw(const char *name, int dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? atoi(s) : dflt;
}

#ifdef MFX_EXPERIMENTS
inline int knob_int(const char *name, int dflt) { return env_int_raw(name, dflt); }
inline float knob_flt(const char *name, float dflt)
{
    const char *s = getenv(name);
    return (s && *s) ? (float)atof(s) : dflt;
}
#else
inline int knob_int(const char *, int dflt) { return dflt; }
inline float knob_flt(const char *, float dflt) { return dflt; }
#endif

} // namespace mfx
