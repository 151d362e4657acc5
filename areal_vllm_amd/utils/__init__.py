"""Runtime services: logging, name resolution, stats, timers, checkpoints, packing, networking."""
