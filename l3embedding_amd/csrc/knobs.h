// knobs.h -- developer switches.
//
// The kernel-variant switches (L3_WG_WINO, L3_BF16_HALO, L3_WGW_UC, ...) exist for A/B measurements and so that the
// tests can run every variant against the oracle; they are NOT configuration.  They are read only when the process sets
// L3_DEBUG_KNOBS=1 (tests/conftest.py does); without it every l3_knob() is "unset" and the library runs its one
// product configuration whatever the environment holds.  Outside this gate the library reads three variables:
// L3_RCCL_LIB (comm.hip: which librccl to dlopen), L3_PROFILE_VERBOSE (engine.hip: print the per-launch table) and
// L3_HOST_WAIT (engine.hip: `spin` keeps HIP's polling host waits, default `block`).
#pragma once
#include <stdlib.h>

namespace l3 {

inline const char* l3_knob(const char* name) {
    const char* on = getenv("L3_DEBUG_KNOBS");
    return on != nullptr && on[0] == '1' ? getenv(name) : nullptr;
}

}  // namespace l3
