// Paged KV block manager with hash-chained prefix caching (host side of the rollout engine).
//
// Pages hold PAGE tokens. A page that is completely filled with prompt/output tokens can be
// published in the prefix cache under hash(prev_page_hash, tokens of the page); later requests
// whose token prefix hashes to the same chain share the page (ref-counted). Unreferenced cached
// pages stay resident until the allocator needs them (LRU). A weight update invalidates the
// cache because cached K/V were produced by older parameters.
//
// A hit is never trusted on the hash alone: every cached page keeps the tokens it holds, a match compares
// them with the request's tokens, and a chain stops being published at the first page whose key is
// already taken by different tokens. By induction over the chain, the K/V of a matched page were computed
// under exactly the request's prefix.
#pragma once
#include <cstdint>
#include <cstring>
#include <deque>
#include <list>
#include <unordered_map>
#include <vector>

namespace areal {

constexpr int PAGE = 32;

inline uint64_t hash_page(uint64_t prev, const int32_t* toks, int n) {
  uint64_t h = prev ^ 0x9E3779B97F4A7C15ull;
  for (int i = 0; i < n; ++i) {
    h ^= (uint64_t)(uint32_t)toks[i] + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
    h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 31;
  }
  return h;
}

class BlockManager {
 public:
  explicit BlockManager(int num_pages, bool enable_prefix_cache = true)
      : num_pages_(num_pages), enable_cache_(enable_prefix_cache), ref_(num_pages, 0),
        page_hash_(num_pages, 0), cached_(num_pages, 0), in_lru_(num_pages, 0),
        page_tokens_(enable_prefix_cache ? (size_t)num_pages * PAGE : 0, 0) {
    // page 0 is reserved as the scratch page that padded / inactive rows write into
    for (int i = num_pages - 1; i >= 1; --i) free_.push_back(i);
  }

  int num_free() const { return (int)free_.size() + (int)lru_.size(); }
  int num_total() const { return num_pages_ - 1; }
  int num_cached() const { return (int)hash_to_page_.size(); }
  int64_t num_collisions() const { return n_collisions_; }
  // test hook: overwrite the tokens remembered for a cached page (simulates a colliding key)
  void corrupt_page_tokens_for_test(int p) {
    if (!page_tokens_.empty()) page_tokens_[(size_t)p * PAGE] ^= 0x5a5a5a;
  }

  // Allocate n fresh pages (evicting unreferenced cached pages if needed). Returns false and
  // allocates nothing when fewer than n are available.
  bool allocate(int n, std::vector<int>& out) {
    if (n > num_free()) return false;
    for (int i = 0; i < n; ++i) {
      int p;
      if (!free_.empty()) {
        p = free_.back();
        free_.pop_back();
      } else {
        p = lru_.front();
        lru_.pop_front();
        in_lru_[p] = 0;
        lru_pos_.erase(p);
        evict(p);
      }
      ref_[p] = 1;
      out.push_back(p);
    }
    return true;
  }

  void add_ref(int p) {
    if (ref_[p] == 0 && in_lru_[p]) {
      lru_.erase(lru_pos_[p]);
      lru_pos_.erase(p);
      in_lru_[p] = 0;
    }
    ++ref_[p];
  }

  void release(int p) {
    if (p <= 0 || ref_[p] <= 0) return;
    if (--ref_[p] == 0) {
      if (cached_[p]) {
        lru_.push_back(p);
        lru_pos_[p] = std::prev(lru_.end());
        in_lru_[p] = 1;
      } else {
        free_.push_back(p);
      }
    }
  }

  void release_all(const std::vector<int>& pages) {
    // release in reverse so that the tail of a chain is evicted before its head
    for (auto it = pages.rbegin(); it != pages.rend(); ++it) release(*it);
  }

  // Longest cached prefix of `toks` in whole pages; never matches the full sequence (at least one
  // token must be left to compute so that the request produces logits). Matched pages get a ref.
  int match_prefix(const int32_t* toks, int n, std::vector<int>& pages) {
    if (!enable_cache_) return 0;
    uint64_t h = 0;
    int matched = 0;
    while (matched + PAGE <= n - 1) {
      h = hash_page(h, toks + matched, PAGE);
      auto it = hash_to_page_.find(h);
      if (it == hash_to_page_.end()) break;
      if (!same_tokens(it->second, toks + matched)) {  // hash collision: not this prefix
        ++n_collisions_;
        break;
      }
      add_ref(it->second);
      pages.push_back(it->second);
      matched += PAGE;
    }
    return matched;
  }

  // Publish the full pages of a sequence (token prefix of length n, page list) in the cache.
  void publish(const int32_t* toks, int n, const std::vector<int>& pages) {
    if (!enable_cache_) return;
    uint64_t h = 0;
    int full = n / PAGE;
    for (int i = 0; i < full && i < (int)pages.size(); ++i) {
      h = hash_page(h, toks + i * PAGE, PAGE);
      int p = pages[i];
      if (cached_[p] && page_hash_[p] == h) continue;  // matched from the cache earlier, or published before
      auto it = hash_to_page_.find(h);
      if (it != hash_to_page_.end()) {
        if (same_tokens(it->second, toks + i * PAGE)) continue;  // another page already serves this prefix
        ++n_collisions_;
        break;  // key taken by other tokens: deeper pages of this chain could not be verified, stop here
      }
      if (cached_[p]) break;  // page is published under another key (cannot happen for owned pages)
      hash_to_page_[h] = p;
      page_hash_[p] = h;
      cached_[p] = 1;
      std::memcpy(&page_tokens_[(size_t)p * PAGE], toks + i * PAGE, PAGE * sizeof(int32_t));
    }
  }

  // Drop every cached prefix (called on weight updates).
  void flush_cache() {
    for (auto& kv : hash_to_page_) cached_[kv.second] = 0;
    hash_to_page_.clear();
    for (int p : lru_) {
      in_lru_[p] = 0;
      free_.push_back(p);
    }
    lru_.clear();
    lru_pos_.clear();
  }

 private:
  bool same_tokens(int p, const int32_t* toks) const {
    return std::memcmp(&page_tokens_[(size_t)p * PAGE], toks, PAGE * sizeof(int32_t)) == 0;
  }
  void evict(int p) {
    if (cached_[p]) {
      hash_to_page_.erase(page_hash_[p]);
      cached_[p] = 0;
    }
  }
  int num_pages_;
  bool enable_cache_;
  std::vector<int> free_;
  std::vector<int> ref_;
  std::vector<uint64_t> page_hash_;
  std::vector<char> cached_, in_lru_;
  std::list<int> lru_;
  std::unordered_map<int, std::list<int>::iterator> lru_pos_;
  std::unordered_map<uint64_t, int> hash_to_page_;
  std::vector<int32_t> page_tokens_;  // [num_pages, PAGE]: tokens of every cached page
  int64_t n_collisions_ = 0;
};

}  // namespace areal
