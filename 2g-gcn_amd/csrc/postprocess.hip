// Inference post-processing on the device, so that a prediction pass hands back labels and
// the segmental metric instead of the (bs, C, T, E) log-probability tensors.
//
// Reference: predict.py:64-70 (torch.repeat_interleave of every output by the downsampling factor + match_shape
// :95-116), :195-201 (np.argmax over the class axis after a D2H copy of every output), and the segmental F1@k metric
// pyrutils/metrics.py:7-81 (run-length encoded segments, greedy IoU matching).
#include "twog_common.h"

namespace {

// labels[b][t'][e] = first argmax_c logp[b][c][min(t' / ds, T - 1)][e]
__global__ __launch_bounds__(256) void predict_labels_kernel(const float* logp, int bs, int C, int T, int E, int ds,
                                                             int T_out, long long* labels) {
    const int64_t n = (int64_t)bs * T_out * E;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int e = (int)(i % E);
        const int64_t bt = i / E;
        const int tp = (int)(bt % T_out), b = (int)(bt / T_out);
        const int t = min(tp / ds, T - 1);
        const float* p = logp + ((int64_t)b * C * T + t) * E + e;
        float best = p[0];
        int arg = 0;
        for (int c = 1; c < C; ++c) {
            const float v = p[(int64_t)c * T * E];
            if (v > best) { best = v; arg = c; }  // strict: ties keep the first index, like np.argmax
        }
        labels[i] = arg;
    }
}

// One thread per sequence. Steps whose target equals the ignore value are dropped from BOTH sequences before the
// run-length encoding (metrics.py:75-77). `used` is a per-sequence scratch row of n_steps bytes.
__global__ __launch_bounds__(64) void f1_at_k_kernel(const long long* y_true, const long long* y_pred, int n_seq,
                                                     int n_steps, int num_classes, double overlap, long long ignore,
                                                     int use_ignore, unsigned char* used, float* f1, float* valid) {
    const int s = blockIdx.x * 64 + threadIdx.x;
    if (s >= n_seq) return;
    const long long* yt = y_true + (int64_t)s * n_steps;
    const long long* yp = y_pred + (int64_t)s * n_steps;
    unsigned char* u = used + (int64_t)s * n_steps;
    // number of target segments in the filtered sequence
    int n_kept = 0, n_tgt = 0;
    long long prev = 0;
    for (int i = 0; i < n_steps; ++i) {
        if (use_ignore && yt[i] == ignore) continue;
        if (n_kept == 0 || yt[i] != prev) { u[n_tgt] = 0; ++n_tgt; }
        prev = yt[i];
        ++n_kept;
    }
    if (n_kept == 0) { f1[s] = 0.f; valid[s] = 0.f; return; }
    double tp = 0.0, fp = 0.0;
    // walk the predicted segments of the filtered sequence
    int pos = 0, i = 0;  // pos: index in the filtered sequence
    while (i < n_steps) {
        if (use_ignore && yt[i] == ignore) { ++i; continue; }
        const long long oid = yp[i];
        const int o0 = pos;
        while (i < n_steps) {  // extend over kept steps with the same predicted label
            if (use_ignore && yt[i] == ignore) { ++i; continue; }
            if (yp[i] != oid) break;
            ++i;
            ++pos;
        }
        const int o1 = pos;
        // IoU against every target segment; first maximum (np.argmax)
        double best = 0.0;
        int best_idx = -1;
        int tpos = 0, seg = -1, t0 = 0;
        long long tid = 0;
        bool open = false;
        for (int j = 0; j <= n_steps; ++j) {
            const bool kept = j < n_steps && !(use_ignore && yt[j] == ignore);
            if (j < n_steps && !kept) continue;
            if (open && (j == n_steps || yt[j] != tid)) {  // close target segment [t0, tpos)
                const double inter = (double)(min(o1, tpos) - max(o0, t0));
                const double uni = (double)(max(o1, tpos) - min(o0, t0));
                const double iou = (inter / uni) * (oid == tid ? 1.0 : 0.0);
                if (best_idx < 0 || iou > best) { best = iou; best_idx = seg; }
                open = false;
            }
            if (j == n_steps) break;
            if (!open) { open = true; tid = yt[j]; t0 = tpos; ++seg; }
            ++tpos;
        }
        if (oid >= num_classes) continue;
        if (best >= overlap && !u[best_idx]) { tp += 1.0; u[best_idx] = 1; }
        else fp += 1.0;
    }
    double n_used = 0.0;
    for (int k = 0; k < n_tgt; ++k) n_used += u[k];
    const double fn = (double)n_tgt - n_used;
    const double precision = tp + fp > 0.0 ? tp / (tp + fp) : 0.0;
    const double recall = tp + fn > 0.0 ? tp / (tp + fn) : 0.0;
    f1[s] = precision + recall > 0.0 ? (float)(2.0 * precision * recall / (precision + recall)) : 0.f;
    valid[s] = 1.f;
}

}  // namespace

extern "C" int twog_predict_labels(const float* logp, int bs, int n_classes, int T, int E, int downsampling, int T_out,
                                   int64_t* labels, void* stream) {
    if (bs < 0 || n_classes < 1 || T < 1 || E < 1 || downsampling < 1 || T_out < 0) return -1;
    const int64_t n = (int64_t)bs * T_out * E;
    if (n == 0) return 0;
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(predict_labels_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, logp, bs, n_classes, T, E,
                       downsampling, T_out, reinterpret_cast<long long*>(labels));
    TWOG_CHECK_LAUNCH();
    return 0;
}

extern "C" int twog_f1_at_k(const int64_t* y_true, const int64_t* y_pred, int n_seq, int n_steps, int num_classes,
                            double overlap, int64_t ignore_value, int use_ignore, unsigned char* scratch, float* f1,
                            float* valid, void* stream) {
    if (n_seq < 0 || n_steps < 0) return -1;
    if (n_seq == 0) return 0;
    hipLaunchKernelGGL(f1_at_k_kernel, dim3((n_seq + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(y_true), reinterpret_cast<const long long*>(y_pred), n_seq,
                       n_steps, num_classes, overlap, (long long)ignore_value, use_ignore, scratch, f1, valid);
    TWOG_CHECK_LAUNCH();
    return 0;
}
