"""MI355X-native asynchronous RL framework for LLMs (GRPO / decoupled PPO / SFT).

Sub-packages: ``api`` (configs, engine/workflow interfaces), ``ops`` (HIP kernels), ``models`` (native
transformer + HF checkpoints), ``parallel`` (process grid, ZeRO sharding, TP/SP/PP), ``engine`` (train,
PPO actor, SFT, remote/local inference clients), ``inference`` (paged-KV continuous-batching server),
``workflow``, ``reward``, ``dataset``, ``launcher``, ``utils``, ``system`` (worker-based runtime).
"""
__version__ = "0.1.0"
