// Bench / test harness, not part of libaos2.so: the step schedule of bench.py -- the reference's Tracking thread (per-frame chain),
// its LocalMapping-side work (the keyframe legs, Optimizer::LocalBundleAdjustment) -- driven by NATIVE threads over the same C-ABI
// calls, in place of five Python threads that hand the interpreter lock to each other between every two calls (System.cc:136-155 runs
// these as std::threads, too).  Python records the calls of every job once (capi.recording(): function address + arguments, all of
// them constants of the run: handles, device pointers, sizes); this library replays the lists:
//
//   step s (pipeline j = s % n_pipes; LocalBA handles take turns, one job per `lba_every` steps -- the windows of that many steps in one call):
//       wait for keyframe job j of the step that used it last (and for the LocalBA job whose handle is due), replay PIPE_WAIT[j]   (main thread)
//       replay PIPE_STEP[j]: the tracking chain of the batch, enqueued                                        (main thread)
//       start KF_JOB[j] on keyframe thread j, every lba_every-th step LBA_JOB[l] on LocalBA thread l          (return at once)
//
// A recorded call is replayed through one function type of 6 + 18 integer and 8 floating-point parameters: on the System V x86-64
// ABI the k-th integer-class argument travels in the k-th integer register (then on the stack, in order) and the k-th
// floating-point one in xmm k whatever the order of the parameters, and a callee that takes fewer simply does not look at the rest.
// (A float is passed as the low 32 bits of its xmm register: the recorder stores the bits accordingly.)  x86-64 Linux only.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#if !defined(__x86_64__) || !defined(__linux__)
#error "host_runner.cpp replays recorded calls through the System V x86-64 calling convention"
#endif

extern "C" {

typedef struct {
    void *fn;
    int32_t n_int, n_fp;
    int64_t iargs[24];
    uint64_t fargs[8];
} aos2_call_t;

enum { AOS2_RUN_PIPE_WAIT = 0, AOS2_RUN_PIPE_STEP = 1, AOS2_RUN_KF_JOB = 2, AOS2_RUN_LBA_JOB = 3, AOS2_RUN_KINDS = 4 };
}

namespace {

typedef int (*replay_fn_t)(long, long, long, long, long, long, double, double, double, double, double, double, double, double, long, long, long,
                           long, long, long, long, long, long, long, long, long, long, long, long, long, long, long);

int invoke(const aos2_call_t &c)
{
    double d[8];
    memcpy(d, c.fargs, sizeof(d));
    const int64_t *a = c.iargs;
    return reinterpret_cast<replay_fn_t>(c.fn)(a[0], a[1], a[2], a[3], a[4], a[5], d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7], a[6], a[7], a[8], a[9],
                                               a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17], a[18], a[19], a[20], a[21], a[22], a[23]);
}

double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Job {   // a worker thread that replays one list when told to
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    const std::vector<aos2_call_t> *list = nullptr;
    int state = 0;   // 0 idle, 1 pending / running
    bool quit = false;
    int status = 0, bad_call = -1;
    std::vector<double> *walls = nullptr;
    std::mutex *walls_m = nullptr;
    std::vector<double> call_sum;   // wall seconds per call index, summed over the runs since reset (diagnostics: which call of a job waits)
    long runs = 0;
    void loop()
    {
        for (;;) {
            const std::vector<aos2_call_t> *l;
            {
                std::unique_lock<std::mutex> lk(m);
                cv.wait(lk, [&] { return quit || (state == 1 && list); });
                if (quit) return;
                l = list;
                list = nullptr;
            }
            const double t0 = now_s();
            int st = 0, bad = -1;
            double per_call[64];
            double tp = t0;
            for (size_t i = 0; i < l->size() && !st; ++i) {
                if ((st = invoke((*l)[i]))) bad = (int)i;
                const double tn = now_s();
                if (i < 64) per_call[i] = tn - tp;
                tp = tn;
            }
            const double dt = tp - t0;
            if (walls) {
                std::lock_guard<std::mutex> g(*walls_m);
                walls->push_back(dt);
                if (!st) {
                    const size_t n = l->size() < 64 ? l->size() : 64;
                    if (call_sum.size() != n) {
                        call_sum.assign(n, 0.0);
                        runs = 0;
                    }
                    for (size_t i = 0; i < n; ++i) call_sum[i] += per_call[i];
                    ++runs;
                }
            }
            std::lock_guard<std::mutex> lk(m);
            if (st && !status) {
                status = st;
                bad_call = bad;
            }
            state = 0;
            cv.notify_all();
        }
    }
    void start(const std::vector<aos2_call_t> *l)
    {
        std::lock_guard<std::mutex> lk(m);
        list = l;
        state = 1;
        cv.notify_all();
    }
    void wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return state == 0; });
    }
};

}  // namespace

extern "C" {

struct aos2_runner {
    int n_pipes = 0, n_lba = 0;
    std::vector<std::vector<aos2_call_t>> lists[AOS2_RUN_KINDS];
    std::vector<Job *> kf, lba;
    std::vector<char> kf_busy, lba_busy;
    double waits[3] = {0, 0, 0};   // LocalBA, keyframe legs, tracking: where the stepping thread waited
    std::vector<double> lba_walls, kf_walls, step_marks;
    std::mutex stats_m;
    int lba_every = 1;     // a LocalBA job is started every `lba_every` steps (it then solves the windows of that many steps in one call)
    int lba_pending = 0;   // steps whose windows wait for the next job
    int lba_next = 0;      // the handle that takes it
    int lba_last = -1;     // the handle of the last job started
    int status = 0;
    char err[256] = {0};
};

aos2_runner *aos2_runner_create(int n_pipes, int n_lba)
{
    if (n_pipes <= 0 || n_lba <= 0) return nullptr;
    aos2_runner *r = new aos2_runner();
    r->n_pipes = n_pipes;
    r->n_lba = n_lba;
    for (int k = 0; k < AOS2_RUN_KINDS; ++k) r->lists[k].resize(k == AOS2_RUN_LBA_JOB ? n_lba : n_pipes);
    auto spawn = [&](std::vector<Job *> &v, int n, std::vector<double> *walls) {
        for (int i = 0; i < n; ++i) {
            Job *j = new Job();
            j->walls = walls;
            j->walls_m = &r->stats_m;
            j->th = std::thread([j] { j->loop(); });
            v.push_back(j);
        }
    };
    spawn(r->kf, n_pipes, &r->kf_walls);
    spawn(r->lba, n_lba, &r->lba_walls);
    r->kf_busy.assign(n_pipes, 0);
    r->lba_busy.assign(n_lba, 0);
    return r;
}

void aos2_runner_destroy(aos2_runner *r)
{
    if (!r) return;
    for (auto *v : {&r->kf, &r->lba})
        for (Job *j : *v) {
            j->wait();
            {
                std::lock_guard<std::mutex> lk(j->m);
                j->quit = true;
                j->cv.notify_all();
            }
            j->th.join();
            delete j;
        }
    delete r;
}

// replaces list (kind, index); n = 0 clears it (the job is then skipped)
int aos2_runner_set_list(aos2_runner *r, int kind, int index, const aos2_call_t *calls, int n)
{
    if (!r || kind < 0 || kind >= AOS2_RUN_KINDS || index < 0 || index >= (int)r->lists[kind].size() || n < 0 || (n > 0 && !calls)) return -1;
    for (int i = 0; i < n; ++i)
        if (!calls[i].fn || calls[i].n_int < 0 || calls[i].n_int > 24 || calls[i].n_fp < 0 || calls[i].n_fp > 8) return -1;
    if (kind == AOS2_RUN_KF_JOB) r->kf[index]->wait();
    if (kind == AOS2_RUN_LBA_JOB) r->lba[index]->wait();
    r->lists[kind][index].assign(calls, calls + n);
    return 0;
}

static int replay(aos2_runner *r, int kind, int index)
{
    const std::vector<aos2_call_t> &l = r->lists[kind][index];
    for (size_t i = 0; i < l.size(); ++i) {
        const int st = invoke(l[i]);
        if (st) {
            if (!r->status) {
                r->status = st;
                snprintf(r->err, sizeof(r->err), "list kind %d index %d call %zu returned %d", kind, index, i, st);
            }
            return st;
        }
    }
    return 0;
}

static void collect(aos2_runner *r, Job *j, const char *what, int index)
{
    if (j->status && !r->status) {
        r->status = j->status;
        snprintf(r->err, sizeof(r->err), "%s job %d call %d returned %d", what, index, j->bad_call, j->status);
    }
}

// the windows of the last `lba_every` steps go to the next LocalBA handle (its previous job waited for first)
static void start_lba(aos2_runner *r)
{
    const int l = r->lba_next;
    if (r->lba_busy[l]) {
        r->lba[l]->wait();
        collect(r, r->lba[l], "LocalBA", l);
        r->lba_busy[l] = 0;
    }
    if (!r->lists[AOS2_RUN_LBA_JOB][l].empty()) {
        r->lba[l]->start(&r->lists[AOS2_RUN_LBA_JOB][l]);
        r->lba_busy[l] = 1;
    }
    r->lba_last = l;
    r->lba_next = (l + 1) % r->n_lba;
    r->lba_pending = 0;
}

// the handle whose LocalBA job was started last (-1: none yet)
int aos2_runner_last_lba(const aos2_runner *r) { return r ? r->lba_last : -1; }

int aos2_runner_set_lba_every(aos2_runner *r, int n)
{
    if (!r || n < 1) return -1;
    r->lba_every = n;
    return 0;
}

int aos2_runner_step(aos2_runner *r, int s)
{
    if (!r || s < 0) return -1;
    const int j = s % r->n_pipes;
    const bool lba_now = r->lba_pending + 1 >= r->lba_every;
    const int l = r->lba_next;
    const double ta = now_s();
    if (lba_now && r->lba_busy[l]) {
        r->lba[l]->wait();
        collect(r, r->lba[l], "LocalBA", l);
        r->lba_busy[l] = 0;
    }
    const double tb = now_s();
    if (r->kf_busy[j]) {
        r->kf[j]->wait();
        collect(r, r->kf[j], "keyframe", j);
        r->kf_busy[j] = 0;
    }
    const double tc = now_s();
    replay(r, AOS2_RUN_PIPE_WAIT, j);
    const double td = now_s();
    r->waits[0] += tb - ta;
    r->waits[1] += tc - tb;
    r->waits[2] += td - tc;
    r->step_marks.push_back(td);
    if (r->status) return r->status;
    replay(r, AOS2_RUN_PIPE_STEP, j);
    if (!r->lists[AOS2_RUN_KF_JOB][j].empty()) {
        r->kf[j]->start(&r->lists[AOS2_RUN_KF_JOB][j]);
        r->kf_busy[j] = 1;
    }
    ++r->lba_pending;
    if (lba_now) start_lba(r);
    return r->status;
}

int aos2_runner_run(aos2_runner *r, int s0, int n)
{
    for (int s = s0; s < s0 + n; ++s)
        if (aos2_runner_step(r, s)) break;
    return r ? r->status : -1;
}

// every job done, every pipeline waited for
int aos2_runner_sync(aos2_runner *r)
{
    if (!r) return -1;
    if (r->lba_pending > 0) start_lba(r);   // (a step count that is no multiple of lba_every: the rest gets its call, too)
    for (int l = 0; l < r->n_lba; ++l)
        if (r->lba_busy[l]) {
            r->lba[l]->wait();
            collect(r, r->lba[l], "LocalBA", l);
            r->lba_busy[l] = 0;
        }
    for (int j = 0; j < r->n_pipes; ++j) {
        if (r->kf_busy[j]) {
            r->kf[j]->wait();
            collect(r, r->kf[j], "keyframe", j);
            r->kf_busy[j] = 0;
        }
        replay(r, AOS2_RUN_PIPE_WAIT, j);
    }
    return r->status;
}

const char *aos2_runner_error(const aos2_runner *r) { return r ? r->err : "no runner"; }

void aos2_runner_reset_stats(aos2_runner *r)
{
    std::lock_guard<std::mutex> g(r->stats_m);
    r->waits[0] = r->waits[1] = r->waits[2] = 0;
    r->lba_walls.clear();
    r->kf_walls.clear();
    r->step_marks.clear();
    for (auto *v : {&r->kf, &r->lba})
        for (Job *j : *v) {
            j->call_sum.clear();
            j->runs = 0;
        }
}

// which: 0 = the three wait sums (seconds), 1 = LocalBA job walls, 2 = keyframe job walls, 3 = step marks, 100 + j = mean wall per call of
// keyframe job j's list, 200 + l = of LocalBA job l's list; returns the count, copies <= cap
int aos2_runner_stats(aos2_runner *r, int which, double *out, int cap)
{
    std::lock_guard<std::mutex> g(r->stats_m);
    const double *src = nullptr;
    int n = 0;
    if (which >= 100) {
        const std::vector<Job *> &v = which >= 200 ? r->lba : r->kf;
        const int i = which >= 200 ? which - 200 : which - 100;
        if (i < 0 || i >= (int)v.size()) return 0;
        const Job *j = v[i];
        n = (int)j->call_sum.size();
        for (int k = 0; k < n && k < cap; ++k) out[k] = j->call_sum[k] / (double)(j->runs > 0 ? j->runs : 1);
        return n;
    }
    if (which == 0) {
        src = r->waits;
        n = 3;
    } else {
        const std::vector<double> &v = which == 1 ? r->lba_walls : which == 2 ? r->kf_walls : r->step_marks;
        src = v.data();
        n = (int)v.size();
    }
    for (int i = 0; i < n && i < cap; ++i) out[i] = src[i];
    return n;
}

}  // extern "C"
