// rl_scheduler.cpp -- TaskScheduler (task_scheduler.rs:48-325) over unit ids.
//
// Same pools, same priority rules and the same get_new_task(completed) protocol as the reference;
// the units themselves (device buffers) stay with the caller, keyed by the ids the reference prints
// (trace_unit.rs:59, plot_unit.rs:37).  Pure host code.
#include <algorithm>
#include <cmath>
#include <deque>
#include <new>

#include <string>

#include "../../include/robigo_luculenta.h"

void rl_internal_set_last_error(const std::string& msg); // rl_api.hip

struct RlScheduler {
    uint32_t traces_completed;                                           // task_scheduler.rs:51
    std::deque<float> performance;                                       // :54
    size_t number_of_trace_units;                                        // :58
    std::deque<uint32_t> available_trace_units, done_trace_units;        // :61-65
    std::deque<uint32_t> available_plot_units, done_plot_units;          // :68-72
    bool gather_unit, tonemap_unit;                                      // :75-78 (Some / None)
    int64_t last_tonemap_time;                                           // :81
    bool image_changed;                                                  // :85
    int64_t tonemap_interval_ms;                                         // :44-46
};

namespace {

void complete_task(RlScheduler* s, const RlTask* t, int64_t now_ms) { // task_scheduler.rs:230-325
    switch (t->kind) {
    case RL_TASK_SLEEP: break;
    case RL_TASK_TRACE:
        s->done_trace_units.push_back(t->unit);
        s->traces_completed += 1;
        break;
    case RL_TASK_PLOT:
        for (uint32_t i = 0; i < t->n_units; ++i) s->available_trace_units.push_back(t->units[i]);
        s->done_plot_units.push_back(t->unit);
        break;
    case RL_TASK_GATHER:
        for (uint32_t i = 0; i < t->n_units; ++i) s->available_plot_units.push_back(t->units[i]);
        s->gather_unit = true;
        s->image_changed = true;
        break;
    case RL_TASK_TONEMAP: {
        s->gather_unit = true;
        s->tonemap_unit = true;
        s->image_changed = false;
        const int64_t render_time = now_ms - s->last_tonemap_time;
        const float batches_per_sec = (float)s->traces_completed * 1000.0f / (float)render_time;
        s->last_tonemap_time = now_ms;
        s->traces_completed = 0;
        s->performance.push_back(batches_per_sec);
        if (s->performance.size() > 512) s->performance.pop_front();
        break;
    }
    default: break;
    }
}

void create_trace_task(RlScheduler* s, RlTask* out) { // task_scheduler.rs:184-190
    out->kind = RL_TASK_TRACE;
    out->unit = s->available_trace_units.front();
    s->available_trace_units.pop_front();
}

void create_plot_task(RlScheduler* s, RlTask* out) { // task_scheduler.rs:192-207
    out->kind = RL_TASK_PLOT;
    out->unit = s->available_plot_units.front();
    s->available_plot_units.pop_front();
    const size_t done = s->done_trace_units.size();
    const size_t n = std::max<size_t>(1, done / 2);
    for (size_t i = 0; i < n && !s->done_trace_units.empty(); ++i) { // pop_front_iter().take(n)
        out->units[out->n_units++] = s->done_trace_units.front();
        s->done_trace_units.pop_front();
    }
}

void create_gather_task(RlScheduler* s, RlTask* out) { // task_scheduler.rs:209-219
    out->kind = RL_TASK_GATHER;
    s->gather_unit = false;
    while (!s->done_plot_units.empty()) {
        out->units[out->n_units++] = s->done_plot_units.front();
        s->done_plot_units.pop_front();
    }
}

void create_tonemap_task(RlScheduler* s, RlTask* out) { // task_scheduler.rs:221-228
    out->kind = RL_TASK_TONEMAP;
    s->gather_unit = false;
    s->tonemap_unit = false;
}

} // namespace

extern "C" {

int rl_scheduler_create(uint32_t concurrency, int64_t tonemap_interval_ms, RlScheduler** out) {
    if (!out) return RL_E_INVALID;
    *out = nullptr;
    const uint32_t n_trace_units = concurrency * 3;                      // task_scheduler.rs:95
    const uint32_t n_plot_units = std::max<uint32_t>(1, concurrency / 2); // :96
    if (concurrency == 0 || n_trace_units > RL_TASK_MAX_UNITS || n_plot_units > RL_TASK_MAX_UNITS) {
        rl_internal_set_last_error("rl_scheduler_create: concurrency must be between 1 and " + std::to_string(RL_TASK_MAX_UNITS / 3) +
                                   " (3 * concurrency trace units have to fit RlTask::units, RL_TASK_MAX_UNITS = " +
                                   std::to_string(RL_TASK_MAX_UNITS) + ")");
        return RL_E_INVALID;
    }
    RlScheduler* s = new (std::nothrow) RlScheduler();
    if (!s) {
        rl_internal_set_last_error("rl_scheduler_create: out of host memory");
        return RL_E_INVALID;
    }
    s->traces_completed = 0;
    s->number_of_trace_units = n_trace_units;
    for (uint32_t i = 0; i < n_trace_units; ++i) s->available_trace_units.push_back(i);
    for (uint32_t i = 0; i < n_plot_units; ++i) s->available_plot_units.push_back(i);
    s->gather_unit = true;
    s->tonemap_unit = true;
    s->last_tonemap_time = 0; // get_time() at construction: the caller's clock starts at 0
    s->image_changed = false;
    s->tonemap_interval_ms = tonemap_interval_ms;
    *out = s;
    return RL_OK;
}

int rl_scheduler_destroy(RlScheduler* s) {
    delete s;
    return RL_OK;
}

int rl_scheduler_get_new_task(RlScheduler* s, const RlTask* completed, int64_t now_ms, RlTask* next) {
    if (!s || !completed || !next) {
        rl_internal_set_last_error("rl_scheduler_get_new_task: null argument");
        return RL_E_INVALID;
    }
    if (completed->n_units > RL_TASK_MAX_UNITS) {
        rl_internal_set_last_error("rl_scheduler_get_new_task: completed task lists more than RL_TASK_MAX_UNITS units");
        return RL_E_INVALID;
    }
    complete_task(s, completed, now_ms); // task_scheduler.rs:129
    next->kind = RL_TASK_SLEEP;
    next->unit = 0;
    next->n_units = 0;

    if (now_ms - s->last_tonemap_time > s->tonemap_interval_ms) { // :133-150
        if (s->image_changed) {
            if (s->gather_unit && s->tonemap_unit) {
                create_tonemap_task(s, next);
                return RL_OK;
            }
        } else if (s->gather_unit && !s->done_plot_units.empty()) {
            create_gather_task(s, next);
            return RL_OK;
        }
    }
    if (s->done_trace_units.size() > s->number_of_trace_units / 2 && !s->available_plot_units.empty()) { // :154-157
        create_plot_task(s, next);
        return RL_OK;
    }
    if (!s->available_trace_units.empty()) { // :160-162
        create_trace_task(s, next);
        return RL_OK;
    }
    if (!s->available_plot_units.empty() && !s->done_trace_units.empty()) { // :166-169
        create_plot_task(s, next);
        return RL_OK;
    }
    if (s->gather_unit && !s->done_plot_units.empty()) { // :174-176
        create_gather_task(s, next);
        return RL_OK;
    }
    return RL_OK; // Task::Sleep, :180
}

int rl_scheduler_performance(RlScheduler* s, float* mean_out, float* stddev_out) { // task_scheduler.rs:317-325
    if (!s) {
        rl_internal_set_last_error("rl_scheduler_performance: null scheduler");
        return RL_E_INVALID;
    }
    const float n = (float)s->performance.size();
    float sum = 0.0f, sq = 0.0f;
    for (float x : s->performance) sum += x;
    for (float x : s->performance) sq += x * x;
    const float mean = sum / n;
    const float variance = sq / n - mean * mean;
    if (mean_out) *mean_out = mean;
    if (stddev_out) *stddev_out = std::sqrt(variance);
    return RL_OK;
}
}
