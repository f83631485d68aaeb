// Training feed: the window gather of TrajectorySlicerDataset.__getitem__ (reference
// envs/dataloaders/trajectory_loader.py:160-197) for a whole batch, on trajectories that live in HBM.
//
//   observations [n_traj][t_max][obs]   actions [n_traj][t_max][act]   (padded trajectories, fp32)
//   slice s = (trajectory slice_traj[s], first row slice_start[s]);  rows [start, start + window)
//   out: observation [batch][window][obs], action [batch][window][act], goal_observation [batch][goal_len][obs]
//
// HBM-bound copy work: one wave per sample (four samples per workgroup), every segment is a contiguous run of the
// source (rows of one trajectory are consecutive), so lanes read and write consecutive 4-byte words; nothing is staged.
#include "common.h"

namespace beso {
namespace {

__global__ __launch_bounds__(256) void gather_windows_kernel(
    const float* __restrict__ observations, const float* __restrict__ actions, const int* __restrict__ seq_len,
    int n_traj, int t_max, int obs, int act, const int* __restrict__ slice_traj, const int* __restrict__ slice_start,
    long long n_slices, const long long* __restrict__ batch_slices, const long long* __restrict__ draws, int window,
    int goal_len, int goal_mode, int min_future_sep, float* __restrict__ obs_out, float* __restrict__ act_out,
    float* __restrict__ goal_out, int batch) {
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6), tid = threadIdx.x & 63;
    if (b >= batch) return;
    const long long sid = batch_slices[b];
    const bool ok = sid >= 0 && sid < n_slices;
    int traj = ok ? slice_traj[sid] : 0, start = ok ? slice_start[sid] : 0;
    const bool in_range = ok && traj >= 0 && traj < n_traj && start >= 0 && start + window <= t_max;
    const size_t row0 = (size_t)traj * t_max + start;
    {
        const int n = window * obs;
        const float* src = observations + row0 * obs;
        float* dst = obs_out + (size_t)b * n;
        for (int i = tid; i < n; i += 64) dst[i] = in_range ? src[i] : 0.f;
    }
    {
        const int n = window * act;
        const float* src = actions + row0 * act;
        float* dst = act_out + (size_t)b * n;
        for (int i = tid; i < n; i += 64) dst[i] = in_range ? src[i] : 0.f;
    }
    if (goal_len > 0) {
        // trajectory_loader.py:168-186: a future sequence of the same trajectory, if one fits behind the window
        const int end = start + window;
        const int lo = end + min_future_sep, hi = (in_range ? seq_len[traj] : 0) - goal_len;
        int g0 = -1;
        if (in_range && lo < hi) {
            if (goal_mode == BESO_GOAL_TAIL) g0 = t_max - goal_len;          // the padded tensor's last rows (:176)
            else if (goal_mode == BESO_GOAL_SEQ_END) g0 = end;               // :178
            else g0 = lo + (int)((unsigned long long)draws[b] % (unsigned long long)(hi - lo));   // :180-182
        }
        const bool have = g0 >= 0 && g0 + goal_len <= t_max;
        const int n = goal_len * obs;
        const float* src = observations + ((size_t)traj * t_max + (have ? g0 : 0)) * obs;
        float* dst = goal_out + (size_t)b * n;
        for (int i = tid; i < n; i += 64) dst[i] = have ? src[i] : 0.f;     // zeros placeholder (:185-186)
    }
}

}  // namespace

hipError_t launch_gather_windows(const float* observations, const float* actions, const int* seq_len, int n_traj,
                                 int t_max, int obs, int act, const int* slice_traj, const int* slice_start,
                                 long long n_slices, const long long* batch_slices, const long long* draws, int batch,
                                 int window, int goal_len, int goal_mode, int min_future_sep, float* obs_out,
                                 float* act_out, float* goal_out, hipStream_t s) {
    (void)hipGetLastError();
    hipLaunchKernelGGL(gather_windows_kernel, dim3((batch + 3) / 4), dim3(256), 0, s, observations, actions, seq_len, n_traj, t_max,
                       obs, act, slice_traj, slice_start, n_slices, batch_slices, draws, window, goal_len, goal_mode,
                       min_future_sep, obs_out, act_out, goal_out, batch);
    return hipGetLastError();
}

}  // namespace beso
