// CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (cgroup v2 cpu.max / v1 cfs_quota).
// Spinning -- or just running -- on more threads than the quota allows gets the whole process throttled for the rest of a
// 100 ms period (the GPU box: 256 CPUs visible, a quota of 16).  Host-only header: the transcript's pools, gkr_verify and the
// R1CS compiler size their thread counts from it.
#pragma once
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <thread>

namespace gkr {

inline int usable_cpus() {
    int hw = (int)std::thread::hardware_concurrency();
    cpu_set_t mask;   // a process pinned to a few cores (taskset, per-rank core binding) must not spin on more threads
    CPU_ZERO(&mask);
    if (sched_getaffinity(0, sizeof mask, &mask) == 0 && CPU_COUNT(&mask) > 0) hw = CPU_COUNT(&mask);
    if (hw < 1) hw = 1;
    double quota = -1, period = -1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        double p = 0;
        if (fscanf(f, "%63s %lf", q, &p) == 2 && strcmp(q, "max") != 0) {
            quota = atof(q);
            period = p;
        }
        fclose(f);
    } else {
        FILE* fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r");
        FILE* fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
        if (fq && fp && fscanf(fq, "%lf", &quota) == 1 && fscanf(fp, "%lf", &period) == 1) {
        } else {
            quota = -1;
        }
        if (fq) fclose(fq);
        if (fp) fclose(fp);
    }
    if (quota > 0 && period > 0) {
        int q = (int)(quota / period);
        if (q < 1) q = 1;
        if (q < hw) hw = q;
    }
    return hw;
}

}  // namespace gkr
