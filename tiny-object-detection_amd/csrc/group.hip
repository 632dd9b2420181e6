// group.hip — frame sharding over the GPUs of one node as a LIBRARY feature.
//
// The reference's caller is one Rust process that owns the frame loop (/root/reference/src/main.rs:63-75 spawns it,
// src/scene.rs:77-92 feeds `classify` one frame at a time); north_star shards camera frames "embarrassingly across the 8 GPUs
// of one node - independent per-GPU batches with weights replicated once via RCCL broadcast over xGMI, no per-step
// collectives". A yh_group is that: one engine handle per device, one host worker thread per handle (so that the H2D copies
// and graph launches of the members are issued concurrently - a single thread would serialise eight ~60 MB pageable copies),
// contiguous frame blocks, results read per global frame index. Nothing here touches another member's data: the only
// collective is the weight replication at load time (yh_group_broadcast_weights; device-to-device copies where members share
// a device, which is how the path is exercised on a one-GPU box).
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "yh_internal.h"

namespace {

thread_local std::string g_group_create_error;

// One worker per member: runs the jobs it is handed, in order, on its own OS thread.
struct Worker {
    std::thread th;
    std::mutex mu;
    std::condition_variable cv;
    std::function<int()> job;
    bool has_job = false, done = true, quit = false;
    int rc = 0;

    void start() {
        th = std::thread([this] {
            for (;;) {
                std::function<int()> j;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [this] { return has_job || quit; });
                    if (quit && !has_job) return;
                    j = std::move(job);
                    has_job = false;
                }
                const int r = j();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    rc = r;
                    done = true;
                }
                cv.notify_all();
            }
        });
    }
    void post(std::function<int()> j) {
        {
            std::lock_guard<std::mutex> lk(mu);
            job = std::move(j);
            has_job = true;
            done = false;
        }
        cv.notify_all();
    }
    int wait() {
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [this] { return done; });
        return rc;
    }
    void stop() {
        {
            std::lock_guard<std::mutex> lk(mu);
            quit = true;
        }
        cv.notify_all();
        if (th.joinable()) th.join();
    }
};

}  // namespace

struct yh_group {
    std::vector<yh_engine*> eng;
    std::vector<int> dev;
    std::vector<Worker*> workers;
    std::vector<int> start, count;   // the partition of the last evaluate: member i owns frames [start[i], start[i] + count[i])
    std::vector<char> warm;          // member i has taken a step before (eager members take their FIRST one on the caller's thread)
    int max_batch = 0, total = 0;
    size_t frame_bytes = 0;
    std::string err, replication = "no weights loaded";
    int fail(int code, const std::string& m) { err = m; return code; }
    int member_fail(int i, int rc) { err = "member " + std::to_string(i) + " (device " + std::to_string(dev[i]) + "): " + yh_last_error(eng[i]); return rc; }
};

extern "C" {

// The caller's current HIP device is the caller's: every engine entry point binds its handle's device on the calling thread
// (include/yolact_hip.h), and a group spans several, so the group's entry points put the caller's device back before they return
// (a torch host that keeps allocating after yh_group_sync must not land on another GPU: ADVICE r3).
struct DeviceGuard {
    int dev = -1;
    DeviceGuard() { if (hipGetDevice(&dev) != hipSuccess) dev = -1; }
    ~DeviceGuard() { if (dev >= 0) hipSetDevice(dev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

const char* yh_group_last_error(const yh_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int yh_group_create(const yh_config* cfg, const int32_t* devices, int32_t n, yh_group** out) {
    DeviceGuard restore_callers_device;
    if (!cfg || !devices || !out || n < 1 || n > 64) { g_group_create_error = "bad argument (1..64 members)"; return YH_EINVAL; }
    *out = nullptr;
    yh_group* g = new yh_group();
    g->max_batch = cfg->max_batch;
    g->frame_bytes = (size_t)cfg->input_size * cfg->input_size * 3;
    for (int i = 0; i < n; ++i) {
        yh_config c = *cfg;
        c.device = devices[i];
        yh_engine* e = nullptr;
        const int rc = yh_create(&c, &e);
        if (rc != YH_OK) {
            g_group_create_error = "member " + std::to_string(i) + " (device " + std::to_string(devices[i]) + "): " + yh_last_error(nullptr);
            yh_group_destroy(g);
            return rc;
        }
        g->eng.push_back(e);
        g->dev.push_back(devices[i]);
    }
    for (int i = 0; i < n; ++i) { g->workers.push_back(new Worker()); g->workers.back()->start(); }
    g->start.assign(n, 0);
    g->count.assign(n, 0);
    g->warm.assign(n, 0);
    *out = g;
    return YH_OK;
}

void yh_group_destroy(yh_group* g) {
    DeviceGuard restore_callers_device;
    if (!g) return;
    for (Worker* w : g->workers) { w->stop(); delete w; }
    for (yh_engine* e : g->eng) yh_destroy(e);
    delete g;
}

int yh_group_size(const yh_group* g) { return g ? (int)g->eng.size() : YH_EINVAL; }
yh_engine* yh_group_member(yh_group* g, int32_t i) { return g && i >= 0 && i < (int)g->eng.size() ? g->eng[i] : nullptr; }
const char* yh_group_weights_replication(const yh_group* g) { return g ? g->replication.c_str() : ""; }

// Member 0 has its weights loaded (yh_load_weights_* / yh_weights_generate on yh_group_member(g, 0)): replicate them.
// Distinct devices: ONE RCCL broadcast of the canonical blob over xGMI (yh_group_broadcast_weights). Members that share a
// device with an earlier member take a device-to-device copy of that member's blob instead (RCCL refuses two ranks on one
// GPU) - the form a one-GPU box can run. If librccl cannot be used, every member on another device receives the blob through
// the host (one D2H + one H2D per member) and the returned string says so.
int yh_group_replicate_weights(yh_group* g) {
    DeviceGuard restore_callers_device;
    if (!g) return YH_EINVAL;
    const int n = (int)g->eng.size();
    const size_t nbytes = yh_weights_nbytes(g->eng[0]);
    if (!yh_weights_device_ptr(g->eng[0])) return g->fail(YH_ESTATE, "load the weights on member 0 first");
    if (n == 1) { g->replication = "single member (no collective)"; return YH_OK; }
    // one representative per distinct device, member 0 first
    std::vector<int> rep, rep_of(n, -1);
    for (int i = 0; i < n; ++i) {
        for (int r : rep) if (g->dev[r] == g->dev[i] && !yh::rccl_shared_device_allowed()) rep_of[i] = r;
        if (rep_of[i] < 0) { rep_of[i] = i; rep.push_back(i); }
    }
    std::string how;
    if (rep.size() > 1) {
        std::vector<yh_engine*> hs;
        for (int r : rep) hs.push_back(g->eng[r]);
        const int rc = yh_group_broadcast_weights(hs.data(), (int)hs.size(), 0);
        if (rc == YH_OK) how = "yh_group_broadcast_weights (RCCL: ncclCommInitAll + grouped ncclBroadcast) over " + std::to_string(rep.size()) + " devices";
        else {
            const std::string why = yh_last_error(g->eng[0]);
            std::vector<uint8_t> host(nbytes);
            if (hipSetDevice(g->dev[0]) != hipSuccess || hipMemcpy(host.data(), yh_weights_device_ptr(g->eng[0]), nbytes, hipMemcpyDeviceToHost) != hipSuccess)
                return g->fail(YH_EHIP, "weight replication through the host failed after: " + why);
            for (size_t k = 1; k < rep.size(); ++k) {
                const int rc2 = yh_load_weights_host(g->eng[rep[k]], host.data(), nbytes);
                if (rc2) return g->member_fail(rep[k], rc2);
            }
            how = "through the host (RCCL path not taken: " + why + ")";
        }
    }
    int shared = 0;
    for (int i = 0; i < n; ++i) {
        if (rep_of[i] == i) continue;
        const int rc = yh_load_weights_device(g->eng[i], yh_weights_device_ptr(g->eng[rep_of[i]]), nbytes);
        if (rc) return g->member_fail(i, rc);
        ++shared;
    }
    if (shared) how += (how.empty() ? "" : "; ") + std::to_string(shared) + " member(s) sharing a device with an earlier member: device-to-device copy";
    g->replication = how;
    return YH_OK;
}

int yh_group_load_weights_host(yh_group* g, const void* blob_host, size_t nbytes) {
    DeviceGuard restore_callers_device;
    if (!g || !blob_host) return YH_EINVAL;
    const int rc = yh_load_weights_host(g->eng[0], blob_host, nbytes);
    if (rc) return g->member_fail(0, rc);
    return yh_group_replicate_weights(g);
}

// fp8 precision: member 0 calibrates on the frames last set on it; every member then runs with member 0's scales (frames of
// one camera stream share a calibration; per-member calibration would make a frame's result depend on which GPU it went to).
int yh_group_fp8_calibrate(yh_group* g) {
    DeviceGuard restore_callers_device;
    if (!g) return YH_EINVAL;
    int rc = yh_fp8_calibrate(g->eng[0]);
    if (rc) return g->member_fail(0, rc);
    const int nl = yh_fp8_layer_count(g->eng[0]);
    for (size_t i = 1; i < g->eng.size(); ++i)
        for (int l = 0; l < nl; ++l) {
            const int nc = yh_fp8_layer_channels(g->eng[0], l);
            if (nc < 1) return g->member_fail(0, YH_EINVAL);
            std::vector<float> sc((size_t)nc, 1.0f);
            if ((rc = yh_fp8_layer_channel_scales(g->eng[0], l, sc.data(), nc))) return g->member_fail(0, rc);
            if ((rc = yh_fp8_set_layer_channel_scales(g->eng[i], l, sc.data(), nc))) return g->member_fail((int)i, rc);
        }
    return YH_OK;
}

static void partition(yh_group* g, int n_frames) {
    const int m = (int)g->eng.size(), base = n_frames / m, rem = n_frames % m;
    for (int i = 0, s = 0; i < m; ++i) {   // contiguous blocks, the first `rem` members one frame more
        g->start[i] = s;
        g->count[i] = base + (i < rem ? 1 : 0);
        s += g->count[i];
    }
    g->total = n_frames;
}

// Once-only work happens HERE, on the caller's thread, one member after the other, while every worker is idle (run_members has
// returned: a worker only ever runs inside it). What that is: the graph capture of a step shape a member has not run yet
// (hipStreamBeginCapture ... hipGraphInstantiate for both of its input buffers, and with it the first resolution of the step's
// kernels on that device). A shape that appears later (another block size) is captured the same way after the members' streams
// have drained. Members that run eagerly (use_graph = 0) have nothing to capture; they take their first step on this thread
// instead (run_members). The worker threads therefore only ever issue steady-state calls - hipSetDevice, event record / wait,
// hipMemcpyAsync, hipGraphLaunch (eager members: kernel launches) - and engine.hip refuses a capture in worker mode. Round 4's
// host segfault (a member capturing beside a neighbour's first pinned allocation, both on worker threads) has no place left to
// happen; its cause inside the runtime stays unproven.
static int ensure_prepared(yh_group* g, int with_tail) {
    const int m = (int)g->eng.size();
    bool drained = false;
    for (int i = 0; i < m; ++i) {
        if (g->count[i] <= 0 || yh::engine_step_prepared(g->eng[i], g->count[i], with_tail)) continue;
        if (!drained) {   // a capture that is needed later than the first step: the group drains first
            for (int k = 0; k < m; ++k) { const int rc = yh_sync(g->eng[k]); if (rc) return g->member_fail(k, rc); }
            drained = true;
        }
        const int rc = yh_prepare(g->eng[i], g->count[i], with_tail);
        if (rc) return g->member_fail(i, rc);
    }
    return YH_OK;
}

static int run_members(yh_group* g, int with_tail, const std::function<int(int)>& job) {
    const int m = (int)g->eng.size();
    int rc0 = ensure_prepared(g, with_tail);
    if (rc0) return rc0;
    std::vector<char> posted((size_t)m, 0);
    int first = YH_OK, who = -1;
    for (int i = 0; i < m; ++i) {   // a member's first step ever, if it runs eagerly: on this thread, before any worker starts
        if (g->count[i] <= 0 || g->warm[i]) continue;
        g->warm[i] = 1;
        if (yh::engine_uses_graph(g->eng[i])) continue;
        const int rc = job(i);
        posted[i] = 2;
        if (rc && !first) { first = rc; who = i; }
    }
    for (int i = 0; i < m; ++i)
        if (g->count[i] > 0 && !posted[i]) {
            posted[i] = 1;
            yh_engine* e = g->eng[i];
            g->workers[i]->post([i, e, &job] {
                yh::WorkerScope in_flight;
                yh::engine_set_worker_mode(e, true);
                const int rc = job(i);
                yh::engine_set_worker_mode(e, false);
                return rc;
            });
        }
    for (int i = 0; i < m; ++i) {
        if (posted[i] != 1) continue;
        const int rc = g->workers[i]->wait();
        if (rc && !first) { first = rc; who = i; }
    }
    return first ? g->member_fail(who, first) : YH_OK;
}

// Captures, on the calling thread and one member at a time, the step every member would run for a call with n_frames frames
// (both input buffers of each block size): what yh_group_evaluate would otherwise do at the first call of that size. A host
// calls it for its block sizes at start-up so that no capture falls into its frame loop. Requires the weights (and, in fp8
// precision, the scales).
int yh_group_prepare(yh_group* g, int32_t n_frames, int32_t with_tail) {
    DeviceGuard restore_callers_device;
    if (!g) return YH_EINVAL;
    const int m = (int)g->eng.size();
    if (n_frames < 1 || (long long)n_frames > (long long)m * g->max_batch) return g->fail(YH_EINVAL, "n_frames must be 1 .. members * max_batch");
    const std::vector<int> s0 = g->start, c0 = g->count;
    const int t0 = g->total;
    partition(g, n_frames);
    const int rc = ensure_prepared(g, with_tail);
    g->start = s0; g->count = c0; g->total = t0;   // (the results of the last evaluate stay addressable)
    return rc;
}

// n_frames u8 RGB frames [n][S][S][3] in HOST memory: member i takes its contiguous block (yh_set_input_u8: the copy runs on
// the member's copy stream underneath its previous step) and enqueues yh_evaluate (with_tail = 1) or yh_invoke. Returns when
// every member has ENQUEUED its step (the caller's frames are free again); the GPUs run on. n_frames <= members * max_batch.
int yh_group_evaluate(yh_group* g, const uint8_t* frames_host, int32_t n_frames, int32_t with_tail) {
    DeviceGuard restore_callers_device;
    if (!g || !frames_host) return YH_EINVAL;
    const int m = (int)g->eng.size();
    if (n_frames < 1 || (long long)n_frames > (long long)m * g->max_batch) return g->fail(YH_EINVAL, "n_frames must be 1 .. members * max_batch");
    yh::TraceRange tr("yh_group_evaluate");
    partition(g, n_frames);
    return run_members(g, with_tail, [&](int i) {
        int rc = yh_set_input_u8(g->eng[i], frames_host + (size_t)g->start[i] * g->frame_bytes, g->count[i]);
        if (rc == YH_OK) rc = with_tail ? yh_evaluate(g->eng[i]) : yh_invoke(g->eng[i]);
        return rc;
    });
}

// The same with frames already resident on each member's OWN device: frames_dev[i] -> counts[i] frames (0 = the member sits
// this step out). The global frame index runs over the members in order.
int yh_group_evaluate_device(yh_group* g, const uint8_t* const* frames_dev, const int32_t* counts, int32_t with_tail) {
    DeviceGuard restore_callers_device;
    if (!g || !frames_dev || !counts) return YH_EINVAL;
    const int m = (int)g->eng.size();
    int total = 0;
    for (int i = 0; i < m; ++i) {
        if (counts[i] < 0 || counts[i] > g->max_batch || (counts[i] > 0 && !frames_dev[i])) return g->fail(YH_EINVAL, "counts[i] must be 0 .. max_batch with a device pointer");
        g->start[i] = total;
        g->count[i] = counts[i];
        total += counts[i];
    }
    if (total < 1) return g->fail(YH_EINVAL, "no frames");
    g->total = total;
    yh::TraceRange tr("yh_group_evaluate_device");
    return run_members(g, with_tail, [&](int i) {
        int rc = yh_set_input_u8_device(g->eng[i], frames_dev[i], g->count[i]);
        if (rc == YH_OK) rc = with_tail ? yh_evaluate(g->eng[i]) : yh_invoke(g->eng[i]);
        return rc;
    });
}

int yh_group_sync(yh_group* g) {
    DeviceGuard restore_callers_device;
    if (!g) return YH_EINVAL;
    for (size_t i = 0; i < g->eng.size(); ++i) {
        const int rc = yh_sync(g->eng[i]);
        if (rc) return g->member_fail((int)i, rc);
    }
    return YH_OK;
}

// Which member holds global frame `frame` of the last evaluate, and at which index of its batch.
int yh_group_frame_owner(const yh_group* g, int32_t frame, int32_t* member, int32_t* local) {
    if (!g || frame < 0 || frame >= g->total) return YH_EINVAL;
    for (size_t i = 0; i < g->eng.size(); ++i)
        if (frame >= g->start[i] && frame < g->start[i] + g->count[i]) {
            if (member) *member = (int)i;
            if (local) *local = frame - g->start[i];
            return YH_OK;
        }
    return YH_EINVAL;
}

int yh_group_read_detections(yh_group* g, int32_t frame, int32_t* count, yh_detection* dets, int32_t dets_capacity, uint8_t* masks, size_t masks_capacity) {
    DeviceGuard restore_callers_device;
    if (!g || !count) return YH_EINVAL;
    int32_t mem = 0, loc = 0;
    if (yh_group_frame_owner(g, frame, &mem, &loc) != YH_OK) return g->fail(YH_EINVAL, "frame out of range of the last yh_group_evaluate");
    const int rc = yh_read_detections(g->eng[mem], loc, count, dets, dets_capacity, masks, masks_capacity);
    return rc ? g->member_fail(mem, rc) : YH_OK;
}

}  // extern "C"
