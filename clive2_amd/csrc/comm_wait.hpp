// comm_wait.hpp -- host-side wait for an enqueued collective, with a deadline.
//
// A collective that a peer never joins does not complete: hipStreamSynchronize() on its stream would
// block for ever, and nothing but an external launcher could end the job.  The library therefore never
// blocks on a collective's stream; it polls the stream together with the communicator's asynchronous
// error state until the work is done, an error shows up, or the deadline passes -- and the caller then
// aborts the communicator (ncclCommAbort) and returns CL2_E_COMM.
//
// The reference has no multi-device code (src/renderer.py:281-291); this belongs to the sample split of
// SURVEY.md 8e.  The function is a template over its two probes so that tests can drive it on a CPU with
// stand-ins for the stream and the communicator (tests/comm_wait_stub.cpp): no HIP or RCCL type appears.
#pragma once
#include <chrono>
#include <thread>

namespace cl2 {

enum WaitResult { WAIT_DONE = 0, WAIT_TIMEOUT = 1, WAIT_ASYNC_ERROR = 2, WAIT_STREAM_ERROR = 3 };

// stream_state(): 0 = all work on the stream has completed, 1 = still running, < 0 = the stream failed.
// async_error(): 0 = the communicator is healthy (or still progressing), otherwise its error code.
// `detail` receives the failing probe's code.
template <class StreamState, class AsyncError>
inline WaitResult wait_collective(StreamState stream_state, AsyncError async_error, double deadline_seconds, int* detail = nullptr) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned spins = 0;
    while (true) {
        const int s = stream_state();
        if (s == 0) return WAIT_DONE;
        if (s < 0) { if (detail) *detail = s; return WAIT_STREAM_ERROR; }
        const int e = async_error();
        if (e != 0) { if (detail) *detail = e; return WAIT_ASYNC_ERROR; }
        const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (waited > deadline_seconds) return WAIT_TIMEOUT;
        // short collectives (the 8-byte barrier of bench.py) finish within the first few polls: spin briefly,
        // then back off so a long wait does not burn a host core
        if (++spins < 2000) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(waited < 0.05 ? 20 : 200));
    }
}

}  // namespace cl2
