// knobs.h -- developer switches.
//
// The kernel-variant switches (L3_WG_WINO, L3_BF16_HALO, L3_WGW_UC, ...) exist for A/B measurements and so that the
// tests can run every variant against the oracle; they are NOT configuration.  They are read only when the process sets
// L3_DEBUG_KNOBS=1 (tests/conftest.py does); without it every l3_knob() is "unset" and the library runs its one
// product configuration whatever the environment holds.  Outside this gate the library reads three variables:
// L3_RCCL_LIB (comm.hip: which librccl to dlopen), L3_PROFILE_VERBOSE (engine.hip: print the per-launch table) and
// L3_HOST_WAIT (stream_wait below: `spin` = hipStreamSynchronize, default `poll`).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

namespace l3 {

inline const char* l3_knob(const char* name) {
    const char* on = getenv("L3_DEBUG_KNOBS");
    return on != nullptr && on[0] == '1' ? getenv(name) : nullptr;
}

// Host wait for a stream.  hipStreamSynchronize spins -- and keeps TWO host threads of the process busy while it does (the caller
// and a runtime thread: 1.7-2.0 cores per rank measured, scripts/probes/cpu_use.py), which is the whole 16-core CPU quota of an
// 8-GPU job on the GPU boxes before its feeds inflate a byte.  The default here asks hipStreamQuery instead and sleeps in between
// (a few back-to-back queries first, then 20 us doubling to 200 us): the caller's thread costs nothing while a 34-ms step runs, the
// wake-up is at most 0.2 ms late.  (hipDeviceScheduleBlockingSync does the same through the completion interrupt, but a process
// under rocprofv3 then never leaves its exit handlers and two processes sharing one GPU hang in l3_destroy -- both measured;
// polling has neither problem.)  L3_HOST_WAIT=spin restores hipStreamSynchronize.
inline hipError_t stream_wait(hipStream_t s) {
    static const bool spin = [] {
        const char* v = getenv("L3_HOST_WAIT");
        return v != nullptr && strcmp(v, "spin") == 0;
    }();
    if (spin) return hipStreamSynchronize(s);
    long ns = 20000;
    for (int tries = 0;; ++tries) {
        const hipError_t r = hipStreamQuery(s);
        if (r != hipErrorNotReady) return r;
        if (tries < 4) continue;
        const struct timespec ts = {0, ns};
        nanosleep(&ts, nullptr);
        if (ns < 200000) ns *= 2;
    }
}

// ... and for an event (the deferred step results)
inline hipError_t event_wait(hipEvent_t ev) {
    static const bool spin = [] {
        const char* v = getenv("L3_HOST_WAIT");
        return v != nullptr && strcmp(v, "spin") == 0;
    }();
    if (spin) return hipEventSynchronize(ev);
    long ns = 20000;
    for (int tries = 0;; ++tries) {
        const hipError_t r = hipEventQuery(ev);
        if (r != hipErrorNotReady) return r;
        if (tries < 4) continue;
        const struct timespec ts = {0, ns};
        nanosleep(&ts, nullptr);
        if (ns < 200000) ns *= 2;
    }
}

}  // namespace l3
