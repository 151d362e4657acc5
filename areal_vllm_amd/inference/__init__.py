"""Native rollout engine: paged KV cache, C++ continuous-batching scheduler, hipGraph decode, HTTP server."""
