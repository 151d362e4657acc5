"""Base services of the worker-based (multi-process, data-flow-graph) runtime: topology, monitoring,
process-wide constants."""
