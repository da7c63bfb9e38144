// CPU stand-in for the wait the library performs after enqueuing a collective (clive2_amd/csrc/comm_wait.hpp,
// used by cl2_reduce_accumulators / cl2_comm_allreduce_f64): the "stream" is complete when every rank has
// arrived, the "communicator" reports an asynchronous error when the file <dir>/async_error exists.
//
//   comm_wait_stub <dir> <rank> <nranks> <join: 0|1> <deadline seconds>
//
// A rank that joins writes <dir>/arrived.<rank> and waits with the product's wait_collective(); on anything
// but WAIT_DONE it "aborts the communicator" (writes <dir>/aborted.<rank>) and exits with 5 (= -CL2_E_COMM),
// which is what the library does with ncclCommAbort.  A rank started with join = 0 skips the collective.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <sys/stat.h>
#include "../clive2_amd/csrc/comm_wait.hpp"

static bool exists(const std::string& p) { struct stat st; return stat(p.c_str(), &st) == 0; }
static void touch(const std::string& p) { FILE* f = std::fopen(p.c_str(), "w"); if (f) std::fclose(f); }

int main(int argc, char** argv) {
    if (argc != 6) return 2;
    const std::string dir = argv[1];
    const int rank = std::atoi(argv[2]), nranks = std::atoi(argv[3]), join = std::atoi(argv[4]);
    const double deadline = std::atof(argv[5]);
    if (!join) { std::printf("rank %d skipped the collective\n", rank); return 0; }
    touch(dir + "/arrived." + std::to_string(rank));
    int detail = 0;
    const cl2::WaitResult res = cl2::wait_collective(
        [&]() -> int {
            for (int k = 0; k < nranks; k++) if (!exists(dir + "/arrived." + std::to_string(k))) return 1;
            return 0;
        },
        [&]() -> int { return exists(dir + "/async_error") ? 6 : 0; },
        deadline, &detail);
    static const char* names[] = {"DONE", "TIMEOUT", "ASYNC_ERROR", "STREAM_ERROR"};
    std::printf("rank %d: %s detail %d\n", rank, names[res], detail);
    if (res == cl2::WAIT_DONE) return 0;
    touch(dir + "/aborted." + std::to_string(rank));
    return 5;
}
