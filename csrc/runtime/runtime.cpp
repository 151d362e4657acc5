// pybind11 module `_runtime`: the C++ host runtime of the framework.
//   * Scheduler / BlockManager  - continuous-batching rollout engine core (scheduler.hpp)
//   * ffd_allocate              - first-fit-decreasing micro-batch allocation
//                                 (semantics of areal/utils/datapack.py:149-184 in the reference)
//   * partition_balanced        - contiguous k-way partition minimising the largest part
//                                 (the numba routine the reference uses for DP sequence balancing)
//   * merge_intervals           - sort + merge of [start, end) intervals
//                                 (csrc/interval_op/interval_op.cpp:1-29 in the reference)
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <numeric>
#include <set>

#include "scheduler.hpp"

namespace py = pybind11;
using namespace areal;

template <typename T>
static py::array_t<T> to_np(const std::vector<T>& v) {
  py::array_t<T> a((py::ssize_t)v.size());
  if (!v.empty()) std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(T));
  return a;
}

static py::dict plan_to_dict(const Plan& p) {
  py::dict d;
  d["kind"] = p.kind;
  d["plan_id"] = p.plan_id;
  d["req_ids"] = to_np(p.req_ids);
  d["input_ids"] = to_np(p.input_ids);
  d["positions"] = to_np(p.positions);
  d["slot_mapping"] = to_np(p.slot_mapping);
  d["temperature"] = to_np(p.temperature);
  d["top_p"] = to_np(p.top_p);
  d["top_k"] = to_np(p.top_k);
  d["frequency_penalty"] = to_np(p.frequency_penalty);
  d["presence_penalty"] = to_np(p.presence_penalty);
  d["penalty_slot"] = to_np(p.penalty_slot);
  if (p.kind == 1) {
    d["cu_q"] = to_np(p.cu_q);
    d["cu_k"] = to_np(p.cu_k);
    d["max_q"] = p.max_q;
    d["max_k"] = p.max_k;
    d["has_prefix"] = p.has_prefix;
    if (p.has_prefix) {
      d["max_pages"] = p.max_pages;
      py::array_t<int32_t> bt({(py::ssize_t)p.req_ids.size(), (py::ssize_t)p.max_pages});
      std::memcpy(bt.mutable_data(), p.block_tables.data(), p.block_tables.size() * sizeof(int32_t));
      d["block_tables"] = bt;
    }
    d["chunk_end"] = to_np(p.chunk_end);
    d["sample"] = to_np(p.sample);
  } else if (p.kind == 2) {
    d["n_steps"] = p.n_steps;
    d["continuation"] = p.continuation;
    d["context_lens"] = to_np(p.context_lens);
    d["max_pages"] = p.max_pages;
    py::array_t<int32_t> bt({(py::ssize_t)p.req_ids.size(), (py::ssize_t)p.max_pages});
    if (!p.block_tables.empty())
      std::memcpy(bt.mutable_data(), p.block_tables.data(), p.block_tables.size() * sizeof(int32_t));
    d["block_tables"] = bt;
    d["remaining"] = to_np(p.remaining);
    d["min_remaining"] = to_np(p.min_remaining);
    py::array_t<int32_t> stops({(py::ssize_t)(p.stop_ids.size() / MAX_DEVICE_STOP_IDS),
                                (py::ssize_t)MAX_DEVICE_STOP_IDS});
    if (!p.stop_ids.empty())
      std::memcpy(stops.mutable_data(), p.stop_ids.data(), p.stop_ids.size() * sizeof(int32_t));
    d["stop_ids"] = stops;
    d["kill"] = to_np(p.kill);
  }
  return d;
}

// Plans stay here from `schedule` until their results are processed (two can be open at once: a decode
// plan and its continuation). `plan_id < 0` addresses the plan scheduled last.
class PyScheduler {
 public:
  explicit PyScheduler(const SchedulerConfig& cfg) : s_(cfg) {}
  int64_t add_request(const std::string& rid, const std::vector<int32_t>& prompt, int max_new_tokens,
                      int min_new_tokens, float temperature, float top_p, int top_k,
                      const std::vector<int32_t>& stop_token_ids,
                      const std::vector<std::vector<int32_t>>& stop_sequences, float frequency_penalty,
                      float presence_penalty, int penalty_slot) {
    SamplingParams sp;
    sp.max_new_tokens = max_new_tokens;
    sp.min_new_tokens = min_new_tokens;
    sp.temperature = temperature;
    sp.top_p = top_p;
    sp.top_k = top_k;
    sp.stop_token_ids = stop_token_ids;
    sp.stop_sequences = stop_sequences;
    sp.frequency_penalty = frequency_penalty;
    sp.presence_penalty = presence_penalty;
    sp.penalty_slot = penalty_slot;
    if ((frequency_penalty != 0.f || presence_penalty != 0.f) && penalty_slot < 0)
      throw std::runtime_error("add_request: a penalty needs a slot of the token-count table");
    return s_.add_request(rid, prompt, sp);
  }
  py::dict remember(Plan&& p) {
    last_ = p.plan_id;
    py::dict d = plan_to_dict(p);
    if (p.kind != 0) open_[p.plan_id] = std::move(p);
    return d;
  }
  py::dict schedule() { return remember(s_.schedule()); }
  py::dict schedule_continuation(int64_t prev_plan_id, int min_live_rows) {
    auto it = open_.find(prev_plan_id);
    if (it == open_.end()) throw std::runtime_error("schedule_continuation: unknown or already processed plan");
    return remember(s_.schedule_continuation(it->second, min_live_rows));
  }
  Plan take(int64_t plan_id) {
    auto it = open_.find(plan_id < 0 ? last_ : plan_id);
    if (it == open_.end()) throw std::runtime_error("process: unknown or already processed plan");
    Plan p = std::move(it->second);
    open_.erase(it);
    return p;
  }
  void process_prefill(py::array_t<int64_t, py::array::c_style | py::array::forcecast> toks,
                       py::array_t<float, py::array::c_style | py::array::forcecast> lps, int64_t plan_id) {
    Plan p = take(plan_id);
    if (p.kind != 1) throw std::runtime_error("process_prefill: not a prefill plan");
    if ((size_t)toks.size() < p.req_ids.size() || (size_t)lps.size() < p.req_ids.size())
      throw std::runtime_error("process_prefill: output shorter than the batch");
    s_.process_prefill(p, toks.data(), lps.data());
  }
  void process_decode(py::array_t<int64_t, py::array::c_style | py::array::forcecast> toks,
                      py::array_t<float, py::array::c_style | py::array::forcecast> lps, int64_t plan_id,
                      bool successor_in_flight) {
    Plan p = take(plan_id);
    if (p.kind != 2) throw std::runtime_error("process_decode: not a decode plan");
    if (toks.ndim() != 2 || toks.shape(0) < p.n_steps || (size_t)toks.shape(1) < p.req_ids.size())
      throw std::runtime_error("process_decode: expected [n_steps, >=batch] outputs");
    s_.process_decode(p, toks.data(), lps.data(), (int)toks.shape(1), successor_in_flight);
  }
  py::list take_finished() {
    py::list out;
    for (auto& f : s_.take_finished()) {
      py::dict d;
      d["rid"] = f.rid;
      d["output_tokens"] = f.output_tokens;
      d["output_logprobs"] = f.output_logprobs;
      d["output_versions"] = f.output_versions;
      d["finish_reason"] = f.finish_reason;
      d["prompt_len"] = f.prompt_len;
      d["penalty_slot"] = f.penalty_slot;
      d["latency"] = f.latency;
      d["ttft"] = f.ttft;
      d["itl"] = f.itl;
      out.append(d);
    }
    return out;
  }
  Scheduler s_;
  std::unordered_map<int64_t, Plan> open_;
  int64_t last_ = 0;
};

// ------------------------------------------------------------------------------ datapack
static std::vector<std::vector<int>> ffd_allocate(const std::vector<int64_t>& values, int64_t capacity,
                                                  int min_groups) {
  const int n = (int)values.size();
  for (auto v : values)
    if (v > capacity) throw std::runtime_error("ffd_allocate: a value exceeds the capacity");
  if (n < min_groups) throw std::runtime_error("ffd_allocate: fewer values than min_groups");
  std::vector<int> order(n);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return values[a] > values[b]; });
  std::vector<std::vector<int>> groups;
  // (sum, group id) ordered by sum: the candidates that can take v are a prefix of this set
  std::set<std::pair<int64_t, int>> by_sum;
  for (int idx : order) {
    const int64_t v = values[idx];
    if ((int)groups.size() < min_groups || by_sum.empty() || by_sum.begin()->first + v > capacity) {
      by_sum.insert({v, (int)groups.size()});
      groups.push_back({idx});
      continue;
    }
    // among groups with room, take the one holding the fewest items (first such on ties)
    auto end = by_sum.upper_bound({capacity - v, n});
    auto best = by_sum.begin();
    for (auto it = by_sum.begin(); it != end; ++it)
      if (groups[it->second].size() < groups[best->second].size()) best = it;
    auto node = *best;
    by_sum.erase(best);
    groups[node.second].push_back(idx);
    by_sum.insert({node.first + v, node.second});
  }
  return groups;
}

// Split `lens` (kept in order) into k contiguous parts minimising the maximum part sum.
// Returns k+1 boundaries. Each part holds at least `min_size` items.
static std::vector<int> partition_balanced(const std::vector<int64_t>& lens, int k, int min_size) {
  const int n = (int)lens.size();
  if (k <= 0 || n < k * std::max(1, min_size)) throw std::runtime_error("partition_balanced: infeasible");
  min_size = std::max(1, min_size);
  std::vector<int64_t> pre(n + 1, 0);
  for (int i = 0; i < n; ++i) pre[i + 1] = pre[i] + lens[i];
  const int64_t INF = INT64_MAX / 4;
  std::vector<std::vector<int64_t>> dp(k + 1, std::vector<int64_t>(n + 1, INF));
  std::vector<std::vector<int>> arg(k + 1, std::vector<int>(n + 1, -1));
  dp[0][0] = 0;
  for (int j = 1; j <= k; ++j)
    for (int i = j * min_size; i <= n - (k - j) * min_size; ++i)
      for (int s = (j - 1) * min_size; s <= i - min_size; ++s) {
        if (dp[j - 1][s] >= INF) continue;
        int64_t c = std::max(dp[j - 1][s], pre[i] - pre[s]);
        if (c < dp[j][i]) { dp[j][i] = c; arg[j][i] = s; }
      }
  std::vector<int> bounds(k + 1);
  int i = n;
  for (int j = k; j >= 0; --j) {
    bounds[j] = i;
    if (j > 0) i = arg[j][i];
  }
  return bounds;
}

static py::array_t<int64_t> merge_intervals(py::array_t<int64_t, py::array::c_style | py::array::forcecast> iv) {
  const py::ssize_t n = iv.size() / 2;
  std::vector<std::pair<int64_t, int64_t>> v(n);
  for (py::ssize_t i = 0; i < n; ++i) v[i] = {iv.data()[2 * i], iv.data()[2 * i + 1]};
  std::sort(v.begin(), v.end());
  std::vector<int64_t> out;
  for (auto& p : v) {
    if (!out.empty() && p.first <= out.back()) out.back() = std::max(out.back(), p.second);
    else { out.push_back(p.first); out.push_back(p.second); }
  }
  return to_np(out);
}

PYBIND11_MODULE(_runtime, m) {
  m.doc() = "C++ host runtime: rollout scheduler, block manager, sequence packing, interval merging";
  m.attr("PAGE_SIZE") = PAGE;
  m.attr("MAX_DEVICE_STOP_IDS") = MAX_DEVICE_STOP_IDS;
  py::class_<SchedulerConfig>(m, "SchedulerConfig")
      .def(py::init<>())
      .def_readwrite("num_pages", &SchedulerConfig::num_pages)
      .def_readwrite("max_running_requests", &SchedulerConfig::max_running_requests)
      .def_readwrite("max_prefill_tokens", &SchedulerConfig::max_prefill_tokens)
      .def_readwrite("context_length", &SchedulerConfig::context_length)
      .def_readwrite("decode_steps", &SchedulerConfig::decode_steps)
      .def_readwrite("enable_prefix_cache", &SchedulerConfig::enable_prefix_cache)
      .def_readwrite("policy", &SchedulerConfig::policy)
      .def_readwrite("schedule_conservativeness", &SchedulerConfig::schedule_conservativeness);
  py::class_<PyScheduler>(m, "Scheduler")
      .def(py::init<const SchedulerConfig&>())
      .def("add_request", &PyScheduler::add_request, py::arg("rid"), py::arg("prompt"), py::arg("max_new_tokens"),
           py::arg("min_new_tokens") = 0, py::arg("temperature") = 1.0f, py::arg("top_p") = 1.0f,
           py::arg("top_k") = 1 << 30, py::arg("stop_token_ids") = std::vector<int32_t>{},
           py::arg("stop_sequences") = std::vector<std::vector<int32_t>>{}, py::arg("frequency_penalty") = 0.f,
           py::arg("presence_penalty") = 0.f, py::arg("penalty_slot") = -1)
      .def("schedule", &PyScheduler::schedule)
      .def("schedule_continuation", &PyScheduler::schedule_continuation, py::arg("prev_plan_id"),
           py::arg("min_live_rows") = 1)
      .def("process_prefill", &PyScheduler::process_prefill, py::arg("tokens"), py::arg("logprobs"),
           py::arg("plan_id") = -1)
      .def("process_decode", &PyScheduler::process_decode, py::arg("tokens"), py::arg("logprobs"),
           py::arg("plan_id") = -1, py::arg("successor_in_flight") = false)
      .def("take_finished", &PyScheduler::take_finished)
      .def("progress",
           [](PyScheduler& s, int64_t id, int from) -> py::object {
             std::vector<int32_t> t, v;
             std::vector<float> l;
             if (!s.s_.progress(id, from, t, l, v)) return py::none();
             return py::make_tuple(t, l, v);
           },
           py::arg("req_id"), py::arg("from_index"))
      .def("pause", [](PyScheduler& s) { s.s_.pause(); })
      .def("resume", [](PyScheduler& s) { s.s_.resume(); })
      .def("paused", [](PyScheduler& s) { return s.s_.paused(); })
      .def("abort_all", [](PyScheduler& s) { return s.s_.abort_all(); })
      .def("abort", [](PyScheduler& s, const std::string& rid) { return s.s_.abort(rid); })
      .def("flush_cache", [](PyScheduler& s) { s.s_.flush_cache(); })
      .def("set_version", [](PyScheduler& s, int v) { s.s_.set_version(v); })
      .def("version", [](PyScheduler& s) { return s.s_.version(); })
      .def("num_waiting", [](PyScheduler& s) { return s.s_.num_waiting(); })
      .def("num_running", [](PyScheduler& s) { return s.s_.num_running(); })
      .def("num_free_pages", [](PyScheduler& s) { return s.s_.num_free_pages(); })
      .def("num_total_pages", [](PyScheduler& s) { return s.s_.num_total_pages(); })
      .def("num_cached_pages", [](PyScheduler& s) { return s.s_.num_cached_pages(); })
      .def("total_generated", [](PyScheduler& s) { return s.s_.total_generated(); })
      .def("total_prompt_tokens", [](PyScheduler& s) { return s.s_.total_prompt_tokens(); })
      .def("total_cached_tokens", [](PyScheduler& s) { return s.s_.total_cached_tokens(); })
      .def("num_preemptions", [](PyScheduler& s) { return s.s_.num_preemptions(); })
      .def("num_prefix_collisions", [](PyScheduler& s) { return s.s_.num_prefix_collisions(); })
      .def("num_deferred_pages", [](PyScheduler& s) { return s.s_.num_deferred_pages(); })
      .def("prefill_possible", [](PyScheduler& s) { return s.s_.prefill_possible(); })
      .def("corrupt_cached_page_for_test",
           [](PyScheduler& s, int page) { s.s_.block_manager().corrupt_page_tokens_for_test(page); })
      .def("has_work", [](PyScheduler& s) { return s.s_.has_work(); });
  m.def("ffd_allocate", &ffd_allocate, py::arg("values"), py::arg("capacity"), py::arg("min_groups"));
  m.def("partition_balanced", &partition_balanced, py::arg("lens"), py::arg("k"), py::arg("min_size") = 1);
  m.def("merge_intervals", &merge_intervals);
}
