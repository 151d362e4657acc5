// Continuous-batching scheduler of the native rollout engine (host side, C++).
//
// Owns request state, the paged block manager and the per-step batch plan. The Python model runner
// asks for a plan (`schedule`), executes it on the GPU (varlen prefill or hipGraph'd multi-step
// decode), and hands sampled tokens back (`process_prefill` / `process_decode`). Semantics follow the
// inference-server behaviour the reference relies on (SGLang flags in areal/api/cli_args.py:398-454;
// sampling fields of areal/api/cli_args.py:81-133; abort-on-weight-update protocol in
// areal/engine/sglang_remote.py:121-309; the slot-refilling generator of
// realhf/impl/model/nn/real_llm_generate.py:670-888):
//   * prefill-first scheduling under a token budget (max_prefill_tokens); a prompt longer than the
//     budget is prefilled in chunks over several plans (only the last chunk samples a token),
//   * decode of all running requests, `decode_steps` tokens per plan. A decode plan can be followed by
//     a CONTINUATION plan that is scheduled before the first one's tokens are known: same rows, device
//     state carried over, only page tables refreshed. Rows whose request finished meanwhile stay in
//     place as dead rows (`kill`), and pages of requests that finish while a successor plan is in
//     flight are released one plan later, when nothing on the device can write to them any more,
//   * stop conditions: stop token ids, stop sequences (token-id sequences matched against the tail of
//     the generated tokens), max_new_tokens, context length; min_new_tokens suppresses stops,
//   * frequency / presence penalties: requests carry a slot of the runner's token-count table,
//   * preemption by recompute when the page pool runs dry,
//   * pause / continue and abort-all for interruptible generation: aborted requests return the tokens
//     generated so far with finish reason "abort", their pages are released and the prefix cache is
//     flushed; the client re-issues prompt+partial output and the KV is recomputed by prefill under
//     the new weights,
//   * every generated token is tagged with the weight version that produced it.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <deque>
#include <memory>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "block_manager.hpp"

namespace areal {

constexpr int MAX_DEVICE_STOP_IDS = 8;  // stop ids per row the device checks itself (the host checks all)

struct SamplingParams {
  int max_new_tokens = 16;
  int min_new_tokens = 0;
  float temperature = 1.0f;  // <= 0: greedy
  float top_p = 1.0f;
  int top_k = 1 << 30;
  std::vector<int32_t> stop_token_ids;
  std::vector<std::vector<int32_t>> stop_sequences;
  float frequency_penalty = 0.f;
  float presence_penalty = 0.f;
  int penalty_slot = -1;  // row of the runner's token-count table (>= 0 iff a penalty is set)
};

enum class ReqState { WAITING, PREFILLING, RUNNING, FINISHED };

struct Request {
  int64_t id = 0;
  std::string rid;
  std::vector<int32_t> tokens;  // prompt followed by generated tokens
  int prompt_len = 0;           // length of the ORIGINAL prompt (generated = tokens.size() - prompt_len)
  SamplingParams sp;
  std::vector<int> pages;
  int num_cached = 0;           // tokens whose K/V are in `pages`
  ReqState state = ReqState::WAITING;
  std::vector<float> logprobs;
  std::vector<int32_t> versions;
  std::string finish_reason;
  double t_arrive = 0, t_first = 0, t_last = 0, t_finish = 0;
  std::vector<float> itl;
  int n_preempted = 0;
  int n_generated() const { return (int)tokens.size() - prompt_len; }
};

struct FinishedRequest {
  std::string rid;
  std::vector<int32_t> output_tokens;
  std::vector<float> output_logprobs;
  std::vector<int32_t> output_versions;
  std::string finish_reason;
  int prompt_len = 0;
  int cached_tokens = 0;
  int penalty_slot = -1;
  double latency = 0, ttft = 0;
  std::vector<float> itl;
};

struct Plan {
  int kind = 0;  // 0 idle, 1 prefill, 2 decode
  int64_t plan_id = 0;
  std::vector<int64_t> req_ids;  // decode continuation: -1 marks a dead row
  // prefill
  std::vector<int32_t> input_ids, positions;
  std::vector<int32_t> cu_q, cu_k;
  std::vector<int64_t> slot_mapping;  // where the new tokens' K/V go
  std::vector<int32_t> chunk_end;     // per request: context length after this plan
  std::vector<int32_t> sample;        // per request: 1 = the chunk reaches the end of the prompt, sample a token
  bool has_prefix = false;
  int max_q = 0, max_k = 0;
  // decode
  int n_steps = 0;
  bool continuation = false;          // rows and device state carried over from the previous decode plan
  std::vector<int32_t> context_lens;
  std::vector<int32_t> block_tables;  // [B, max_pages]; prefill plans carry it when a row has context in the cache
  int max_pages = 0;
  std::vector<int32_t> remaining;      // tokens the row may still generate (max_new_tokens and context cap)
  std::vector<int32_t> min_remaining;  // tokens until stop tokens count (min_new_tokens)
  std::vector<int32_t> stop_ids;       // [B, MAX_DEVICE_STOP_IDS], -1 padded
  std::vector<int32_t> kill;           // continuation: 1 = the row's request is gone, deactivate it
  // sampling (both kinds)
  std::vector<float> temperature, top_p, frequency_penalty, presence_penalty;
  std::vector<int32_t> top_k, penalty_slot;
};

struct SchedulerConfig {
  int num_pages = 1024;
  int max_running_requests = 256;
  int max_prefill_tokens = 32768;
  int context_length = 32768;
  int decode_steps = 1;
  bool enable_prefix_cache = true;
  std::string policy = "fcfs";  // "fcfs" | "lpm"
  float schedule_conservativeness = 1.0f;
};

inline double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

class Scheduler {
 public:
  explicit Scheduler(const SchedulerConfig& cfg)
      : cfg_(cfg), bm_(cfg.num_pages, cfg.enable_prefix_cache) {}

  int64_t add_request(const std::string& rid, const std::vector<int32_t>& prompt, const SamplingParams& sp) {
    auto r = std::make_unique<Request>();
    r->id = next_id_++;
    r->rid = rid;
    r->tokens = prompt;
    r->prompt_len = (int)prompt.size();
    r->sp = sp;
    r->t_arrive = now_s();
    int64_t id = r->id;
    if (prompt.empty() || (int)prompt.size() >= cfg_.context_length || sp.max_new_tokens <= 0) {
      r->finish_reason = "length";
      finish(*r);
      return id;
    }
    waiting_.push_back(id);
    reqs_[id] = std::move(r);
    return id;
  }

  void pause() { paused_ = true; }
  void resume() { paused_ = false; }
  bool paused() const { return paused_; }
  void set_version(int v) { version_ = v; }
  int version() const { return version_; }

  // Abort everything in flight (weight update). The caller must have processed every plan it launched:
  // pages are released at once. Returns the number of aborted requests.
  int abort_all() {
    release_deferred();
    int n = 0;
    std::vector<int64_t> ids;
    for (auto id : running_) ids.push_back(id);
    for (auto id : prefilling_) ids.push_back(id);
    for (auto id : waiting_) ids.push_back(id);
    running_.clear();
    prefilling_.clear();
    waiting_.clear();
    for (auto id : ids) {
      auto it = reqs_.find(id);
      if (it == reqs_.end()) continue;
      it->second->finish_reason = "abort";
      release_pages(*it->second, false);
      finish(*it->second);
      reqs_.erase(it);
      ++n;
    }
    bm_.flush_cache();
    return n;
  }

  // Streaming: the tokens request `id` has generated beyond its first `from` (with log-probs and weight versions).
  // False when the request is no longer live (its complete output is in the finished list).
  bool progress(int64_t id, int from, std::vector<int32_t>& toks, std::vector<float>& lps,
                std::vector<int32_t>& vers) const {
    auto it = reqs_.find(id);
    if (it == reqs_.end()) return false;
    const Request& r = *it->second;
    const int n = r.n_generated();
    for (int i = std::max(0, from); i < n; ++i) {
      toks.push_back(r.tokens[r.prompt_len + i]);
      lps.push_back(i < (int)r.logprobs.size() ? r.logprobs[i] : 0.f);
      vers.push_back(i < (int)r.versions.size() ? r.versions[i] : version_);
    }
    return true;
  }

  bool abort(const std::string& rid) {
    release_deferred();
    for (auto& kv : reqs_) {
      if (kv.second->rid == rid) {
        int64_t id = kv.first;
        running_.erase(std::remove(running_.begin(), running_.end(), id), running_.end());
        prefilling_.erase(std::remove(prefilling_.begin(), prefilling_.end(), id), prefilling_.end());
        waiting_.erase(std::remove(waiting_.begin(), waiting_.end(), id), waiting_.end());
        kv.second->finish_reason = "abort";
        release_pages(*kv.second, false);
        finish(*kv.second);
        reqs_.erase(id);
        return true;
      }
    }
    return false;
  }

  void flush_cache() { bm_.flush_cache(); }

  // ------------------------------------------------------------------------------ planning
  bool prefill_possible() const {
    return !paused_ && (!prefilling_.empty() ||
                        (!waiting_.empty() &&
                         (int)(running_.size() + prefilling_.size()) < cfg_.max_running_requests));
  }

  Plan schedule() {
    Plan p;
    p.plan_id = next_plan_++;
    if (prefill_possible()) {
      plan_prefill(p);
      if (p.kind == 1) return p;
    }
    if (!running_.empty() && !paused_) plan_decode(p);
    return p;
  }

  // A decode plan that follows `prev` (a decode plan whose results are NOT processed yet) on the same
  // rows: device-side state (tokens, positions, slots, activity) carries over, only page tables and the
  // kill mask travel. Returns kind 0 when a normal plan has to be made instead: a prefill is due, the
  // batch would fit a much smaller bucket, or pages for the extra steps cannot be had without preemption.
  Plan schedule_continuation(const Plan& prev, int min_live_rows) {
    Plan p;
    p.plan_id = next_plan_++;
    if (prev.kind != 2 || paused_ || prefill_possible()) return p;
    const int k = std::max(1, cfg_.decode_steps);
    int live = 0, need_total = 0;
    for (auto id : prev.req_ids) {
      auto it = id < 0 ? reqs_.end() : reqs_.find(id);
      if (it == reqs_.end() || it->second->state != ReqState::RUNNING) continue;
      ++live;
      const Request& r = *it->second;
      // positions written by `prev` (not processed yet) and by this plan
      int last_pos = std::min((int)r.tokens.size() - 1 + (2 * k - 1), cfg_.context_length - 1);
      need_total += std::max(0, pages_for(last_pos + 1) - (int)r.pages.size());
    }
    if (live == 0 || live < min_live_rows || live != (int)running_.size() || need_total > bm_.num_free()) return p;
    p.kind = 2;
    p.n_steps = k;
    p.continuation = true;
    int max_pages = 0;
    for (auto id : prev.req_ids) {
      auto it = id < 0 ? reqs_.end() : reqs_.find(id);
      if (it == reqs_.end() || it->second->state != ReqState::RUNNING) continue;
      Request& r = *it->second;
      int last_pos = std::min((int)r.tokens.size() - 1 + (2 * k - 1), cfg_.context_length - 1);
      int need = pages_for(last_pos + 1) - (int)r.pages.size();
      if (need > 0) bm_.allocate(need, r.pages);
      max_pages = std::max(max_pages, (int)r.pages.size());
    }
    p.max_pages = max_pages;
    const size_t B = prev.req_ids.size();
    p.block_tables.assign(B * (size_t)max_pages, 0);
    p.req_ids.assign(B, -1);
    p.kill.assign(B, 1);
    for (size_t row = 0; row < B; ++row) {
      int64_t id = prev.req_ids[row];
      auto it = id < 0 ? reqs_.end() : reqs_.find(id);
      if (it == reqs_.end() || it->second->state != ReqState::RUNNING) continue;
      Request& r = *it->second;
      p.req_ids[row] = id;
      p.kill[row] = 0;
      std::copy(r.pages.begin(), r.pages.end(), p.block_tables.begin() + row * max_pages);
    }
    return p;
  }

  // ---------------------------------------------------------------------- result handling
  void process_prefill(const Plan& p, const int64_t* tokens, const float* logprobs) {
    release_deferred();
    double t = now_s();
    for (size_t i = 0; i < p.req_ids.size(); ++i) {
      auto it = reqs_.find(p.req_ids[i]);
      if (it == reqs_.end()) continue;
      Request& r = *it->second;
      r.num_cached = p.chunk_end[i];
      if (!p.sample[i]) continue;  // more chunks of this prompt to come
      bm_.publish(r.tokens.data(), r.num_cached, r.pages);
      append_token(r, (int32_t)tokens[i], logprobs[i], t);
    }
    sweep_finished(false);
  }

  // tokens / logprobs are [n_steps, B_padded] row-major with row stride `stride`. `successor_in_flight`:
  // a continuation of this plan is already running on the device.
  void process_decode(const Plan& p, const int64_t* tokens, const float* logprobs, int stride,
                      bool successor_in_flight = false) {
    release_deferred();  // what the plan before this one left behind: nothing can touch those pages now
    double t = now_s();
    for (size_t i = 0; i < p.req_ids.size(); ++i) {
      auto it = p.req_ids[i] < 0 ? reqs_.end() : reqs_.find(p.req_ids[i]);
      if (it == reqs_.end()) continue;
      Request& r = *it->second;
      for (int s = 0; s < p.n_steps; ++s) {
        if (r.state == ReqState::FINISHED) break;
        r.num_cached = (int)r.tokens.size();  // the consumed token's K/V is now cached
        append_token(r, (int32_t)tokens[(size_t)s * stride + i], logprobs[(size_t)s * stride + i], t);
      }
    }
    sweep_finished(successor_in_flight);
  }

  std::vector<FinishedRequest> take_finished() {
    std::vector<FinishedRequest> out;
    out.swap(finished_);
    return out;
  }

  int num_waiting() const { return (int)waiting_.size(); }
  int num_running() const { return (int)(running_.size() + prefilling_.size()); }
  int num_free_pages() const { return bm_.num_free(); }
  int num_total_pages() const { return bm_.num_total(); }
  int num_cached_pages() const { return bm_.num_cached(); }
  int64_t total_generated() const { return total_generated_; }
  int64_t total_prompt_tokens() const { return total_prompt_; }
  int64_t total_cached_tokens() const { return total_cached_; }
  int64_t num_preemptions() const { return n_preempt_; }
  int64_t num_prefix_collisions() const { return bm_.num_collisions(); }
  int num_deferred_pages() const {
    int n = 0;
    for (auto& z : deferred_) n += (int)z->pages.size();
    return n;
  }
  std::vector<int> deferred_pages() const {
    std::vector<int> out;
    for (auto& z : deferred_) out.insert(out.end(), z->pages.begin(), z->pages.end());
    return out;
  }
  bool has_work() const { return !waiting_.empty() || !running_.empty() || !prefilling_.empty(); }
  BlockManager& block_manager() { return bm_; }

 private:
  static int pages_for(int ntok) { return (ntok + PAGE - 1) / PAGE; }

  void fill_sampling(Plan& p, const Request& r, bool sampled = true) {
    p.temperature.push_back(r.sp.temperature);
    p.top_p.push_back(r.sp.top_p);
    p.top_k.push_back(r.sp.top_k);
    p.frequency_penalty.push_back(r.sp.frequency_penalty);
    p.presence_penalty.push_back(r.sp.presence_penalty);
    p.penalty_slot.push_back(sampled ? r.sp.penalty_slot : -1);
  }

  // Rows of a prefill plan for request r: tokens [r.num_cached, end).
  void emit_prefill_rows(Plan& p, Request& r, int end) {
    const int n = (int)r.tokens.size();
    const int start = r.num_cached;
    for (int t = start; t < end; ++t) {
      p.input_ids.push_back(r.tokens[t]);
      p.positions.push_back(t);
      p.slot_mapping.push_back((int64_t)r.pages[t / PAGE] * PAGE + t % PAGE);
    }
    p.cu_q.push_back(p.cu_q.back() + (end - start));
    p.cu_k.push_back(p.cu_k.back() + end);
    p.max_q = std::max(p.max_q, end - start);
    p.max_k = std::max(p.max_k, end);
    p.req_ids.push_back(r.id);
    p.chunk_end.push_back(end);
    p.sample.push_back(end == n ? 1 : 0);
    if (start > 0) p.has_prefix = true;
    fill_sampling(p, r, end == n);
  }

  void plan_prefill(Plan& p) {
    int budget = std::max(1, cfg_.max_prefill_tokens);
    int reserve = (int)(running_.size() * cfg_.schedule_conservativeness);
    p.cu_q.push_back(0);
    p.cu_k.push_back(0);
    std::vector<std::pair<int64_t, int>> rows;  // (request, chunk end)
    // 1. prompts that are part-way through their prefill go first (they already own their pages)
    for (auto id : prefilling_) {
      if (budget <= 0) break;
      Request& r = *reqs_[id];
      int end = std::min((int)r.tokens.size(), r.num_cached + budget);
      budget -= end - r.num_cached;
      rows.emplace_back(id, end);
    }
    // 2. new admissions
    if (cfg_.policy == "lpm" && waiting_.size() > 1) sort_waiting_lpm();
    while (budget > 0 && !waiting_.empty() &&
           (int)(running_.size() + prefilling_.size()) < cfg_.max_running_requests) {
      int64_t id = waiting_.front();
      Request& r = *reqs_[id];
      int n = (int)r.tokens.size();
      std::vector<int> pages;
      int matched = bm_.match_prefix(r.tokens.data(), n, pages);
      // pages for the whole context plus the first generated token
      int need = pages_for(n + 1) - (int)pages.size();
      if (need > bm_.num_free() - reserve) {
        bm_.release_all(pages);
        const bool alone = rows.empty() && running_.empty() && prefilling_.empty();
        if (alone && need <= bm_.num_free()) {
          // nothing else is running: ignore the reserve so a single large request can proceed
        } else if (alone) {
          // can never fit in the page pool: fail the request instead of stalling the queue
          waiting_.pop_front();
          r.finish_reason = "length";
          finish(r);
          reqs_.erase(id);
          continue;
        } else {
          break;
        }
        matched = bm_.match_prefix(r.tokens.data(), n, pages);
        need = pages_for(n + 1) - (int)pages.size();
      }
      if (!bm_.allocate(need, pages)) {
        bm_.release_all(pages);
        break;
      }
      waiting_.pop_front();
      r.pages = std::move(pages);
      r.num_cached = matched;
      r.state = ReqState::PREFILLING;
      prefilling_.push_back(id);
      total_prompt_ += n - matched;
      total_cached_ += matched;
      int end = std::min(n, matched + budget);
      budget -= end - matched;
      rows.emplace_back(id, end);
    }
    if (rows.empty()) return;
    p.kind = 1;
    for (auto& [id, end] : rows) emit_prefill_rows(p, *reqs_[id], end);
    if (p.has_prefix) {
      // some row attends to context that is in the cache already (a reused prefix, earlier chunks): the attention
      // kernel reads every row's context through its page table
      p.max_pages = pages_for(p.max_k);
      p.block_tables.assign(rows.size() * (size_t)p.max_pages, 0);
      size_t row = 0;
      for (auto& [id, end] : rows) {
        Request& r = *reqs_[id];
        std::copy(r.pages.begin(), r.pages.begin() + pages_for(end), p.block_tables.begin() + row * p.max_pages);
        ++row;
      }
    }
    // rows whose last chunk is in this plan start decoding after it
    for (auto& [id, end] : rows) {
      Request& r = *reqs_[id];
      if (end == (int)r.tokens.size()) {
        prefilling_.erase(std::remove(prefilling_.begin(), prefilling_.end(), id), prefilling_.end());
        r.state = ReqState::RUNNING;
        running_.push_back(id);
      }
    }
  }

  void plan_decode(Plan& p) {
    int k = std::max(1, cfg_.decode_steps);
    // make sure every running request owns pages for the next k positions; preempt from the back
    size_t i = 0;
    while (i < running_.size()) {
      Request& r = *reqs_[running_[i]];
      int last_pos = (int)r.tokens.size() - 1 + (k - 1);
      last_pos = std::min(last_pos, cfg_.context_length - 1);
      int need = pages_for(last_pos + 1) - (int)r.pages.size();
      if (need > 0 && !bm_.allocate(need, r.pages)) {
        if (running_.size() == 1) break;  // cannot preempt the only request; it will hit the length cap
        preempt_last();
        if (i >= running_.size()) break;
        continue;
      }
      ++i;
    }
    if (running_.empty()) return;
    p.kind = 2;
    p.n_steps = k;
    int max_pages = 0;
    for (auto id : running_) max_pages = std::max(max_pages, (int)reqs_[id]->pages.size());
    p.max_pages = max_pages;
    p.block_tables.assign(running_.size() * (size_t)max_pages, 0);
    p.stop_ids.assign(running_.size() * (size_t)MAX_DEVICE_STOP_IDS, -1);
    size_t row = 0;
    for (auto id : running_) {
      Request& r = *reqs_[id];
      int pos = (int)r.tokens.size() - 1;
      p.req_ids.push_back(id);
      p.input_ids.push_back(r.tokens.back());
      p.positions.push_back(pos);
      p.context_lens.push_back(pos + 1);
      p.slot_mapping.push_back((int64_t)r.pages[pos / PAGE] * PAGE + pos % PAGE);
      std::copy(r.pages.begin(), r.pages.end(), p.block_tables.begin() + row * max_pages);
      p.remaining.push_back(std::max(
          1, std::min(r.sp.max_new_tokens - r.n_generated(), cfg_.context_length - (int)r.tokens.size())));
      p.min_remaining.push_back(std::max(0, r.sp.min_new_tokens - r.n_generated()));
      for (size_t j = 0; j < r.sp.stop_token_ids.size() && j < (size_t)MAX_DEVICE_STOP_IDS; ++j)
        p.stop_ids[row * MAX_DEVICE_STOP_IDS + j] = r.sp.stop_token_ids[j];
      p.kill.push_back(0);
      fill_sampling(p, r);
      ++row;
    }
  }

  void preempt_last() {
    int64_t id = running_.back();
    running_.pop_back();
    Request& r = *reqs_[id];
    release_pages(r, false);
    r.state = ReqState::WAITING;
    r.num_cached = 0;
    r.n_preempted++;
    ++n_preempt_;
    waiting_.push_front(id);
  }

  void sort_waiting_lpm() {
    // longest-prefix-match first among the head of the queue (bounded work per step)
    size_t n = std::min<size_t>(waiting_.size(), 128);
    std::vector<std::pair<int, int64_t>> keyed;
    for (size_t i = 0; i < n; ++i) {
      Request& r = *reqs_[waiting_[i]];
      std::vector<int> pages;
      int m = bm_.match_prefix(r.tokens.data(), (int)r.tokens.size(), pages);
      bm_.release_all(pages);
      keyed.emplace_back(-m, waiting_[i]);
    }
    std::stable_sort(keyed.begin(), keyed.end(),
                     [](const auto& a, const auto& b) { return a.first < b.first; });
    for (size_t i = 0; i < n; ++i) waiting_[i] = keyed[i].second;
  }

  // does the tail of the generated tokens equal one of the request's stop sequences?
  static bool ends_with_stop_sequence(const Request& r) {
    const int gen = r.n_generated();
    for (auto& seq : r.sp.stop_sequences) {
      const int L = (int)seq.size();
      if (L == 0 || L > gen) continue;
      if (std::equal(seq.begin(), seq.end(), r.tokens.end() - L)) return true;
    }
    return false;
  }

  void append_token(Request& r, int32_t tok, float lp, double t) {
    r.tokens.push_back(tok);
    r.logprobs.push_back(lp);
    r.versions.push_back(version_);
    ++total_generated_;
    if (r.t_first == 0) r.t_first = t;
    else r.itl.push_back((float)(t - r.t_last));
    r.t_last = t;
    int gen = r.n_generated();
    bool stop = false;
    if (gen >= r.sp.min_new_tokens) {
      for (auto s : r.sp.stop_token_ids)
        if (s == tok) { stop = true; break; }
      if (!stop && !r.sp.stop_sequences.empty()) stop = ends_with_stop_sequence(r);
    }
    if (stop) {
      r.finish_reason = "stop";
      r.state = ReqState::FINISHED;
    } else if (gen >= r.sp.max_new_tokens || (int)r.tokens.size() >= cfg_.context_length) {
      r.finish_reason = "length";
      r.state = ReqState::FINISHED;
    }
  }

  // Finished requests leave the running list and are delivered at once. Their pages go back to the pool
  // now, or - `defer` - after the next processed plan: a successor plan already running on the device
  // still writes K/V for those rows.
  void sweep_finished(bool defer) {
    std::vector<int64_t> keep;
    for (auto id : running_) {
      auto it = reqs_.find(id);
      Request& r = *it->second;
      if (r.state == ReqState::FINISHED) {
        finish(r);
        if (defer) {
          deferred_.push_back(std::move(it->second));
        } else {
          release_pages(r, true);
        }
        reqs_.erase(it);
      } else {
        keep.push_back(id);
      }
    }
    running_.swap(keep);
  }

  void release_deferred() {
    for (auto& z : deferred_) release_pages(*z, true);
    deferred_.clear();
  }

  void release_pages(Request& r, bool publish) {
    if (publish && r.num_cached > 0) bm_.publish(r.tokens.data(), r.num_cached, r.pages);
    bm_.release_all(r.pages);
    r.pages.clear();
  }

  void finish(Request& r) {
    FinishedRequest f;
    r.t_finish = now_s();
    f.rid = r.rid;
    f.output_tokens.assign(r.tokens.begin() + r.prompt_len, r.tokens.end());
    f.output_logprobs = r.logprobs;
    f.output_versions = r.versions;
    f.finish_reason = r.finish_reason;
    f.prompt_len = r.prompt_len;
    f.penalty_slot = r.sp.penalty_slot;
    f.latency = r.t_finish - r.t_arrive;
    f.ttft = r.t_first > 0 ? r.t_first - r.t_arrive : 0;
    f.itl = r.itl;
    r.state = ReqState::FINISHED;
    finished_.push_back(std::move(f));
  }

  SchedulerConfig cfg_;
  BlockManager bm_;
  std::unordered_map<int64_t, std::unique_ptr<Request>> reqs_;
  std::deque<int64_t> waiting_;
  std::vector<int64_t> prefilling_;  // admitted, prompt not completely prefilled yet (own their pages)
  std::vector<int64_t> running_;
  std::vector<std::unique_ptr<Request>> deferred_;  // finished, pages still possibly written by the device
  std::vector<FinishedRequest> finished_;
  int64_t next_id_ = 1, next_plan_ = 1;
  bool paused_ = false;
  int version_ = 0;
  int64_t total_generated_ = 0, total_prompt_ = 0, total_cached_ = 0, n_preempt_ = 0;
};

}  // namespace areal
