// Stress test of HostPool (bulletproofs-amcl_amd/csrc/bp_hostpool.hpp): alternating run(4) / run(8) with tiny jobs, the pattern
// of an MSM (4 tail chains) followed by a paired MSM (8 jobs).  Every job of every run must execute exactly once and run() must not
// return before its jobs have finished (the jobs write to the caller's stack frame).  Built with -fsanitize=thread by
// tests/test_host_cpu.py; exits 0 on success.
#include <cstdio>
#include <cstdlib>
#include "../../bulletproofs-amcl_amd/csrc/bp_hostpool.hpp"

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    HostPool pool;
    long bad = 0;
    for (int it = 0; it < iters; it++) {
        const int n = (it & 1) ? 8 : 4;
        int hits[8] = {0, 0, 0, 0, 0, 0, 0, 0};             // on this frame: a job that runs after run() returned corrupts the next frame's array
        std::function<void(int)> job = [&](int j) { hits[j]++; };
        pool.run(n, job, n - 1);
        for (int j = 0; j < 8; j++) if (hits[j] != (j < n ? 1 : 0)) bad++;
    }
    if (bad) { fprintf(stderr, "hostpool_stress: %ld jobs ran a wrong number of times\n", bad); return 1; }
    printf("hostpool_stress ok: %d runs\n", iters);
    return 0;
}
