"""Keys of the name-resolve store.

Everything a trial registers lives under ``{user}/{experiment}/{trial}/...``, so one `clear_subtree` of the
trial root forgets a whole run; the trial registry lives under ``trial_registry/{user}/...``. The key SPELLINGS
are the cross-process contract (launchers, trainers, servers and workers each build them on their own) and
match ``areal/utils/names.py:11-28`` / ``realhf/base/names.py`` of the reference.
"""
import getpass

USER_NAMESPACE = getpass.getuser()


def _under_trial(experiment_name, trial_name, *parts, trailing_slash: bool = False) -> str:
    key = "/".join(str(p) for p in (USER_NAMESPACE, experiment_name, trial_name, *parts))
    return key + "/" if trailing_slash else key


# ---- registry of trials
def registry_root(user):
    return f"trial_registry/{user}"


def trial_registry(experiment_name, trial_name):
    return f"{registry_root(USER_NAMESPACE)}/{experiment_name}/{trial_name}"


def trial_root(experiment_name, trial_name):
    return _under_trial(experiment_name, trial_name)


# ---- rollout servers and weight versions
def gen_servers(experiment_name, trial_name):
    return _under_trial(experiment_name, trial_name, "gen_servers")


def gen_server_manager(experiment_name, trial_name):
    return _under_trial(experiment_name, trial_name, "gen_server_mgr")


def update_weights_from_disk(experiment_name, trial_name, model_version):
    return _under_trial(experiment_name, trial_name, "update_weights_from_disk", model_version)


def update_weights_from_ipc(experiment_name, trial_name, model_version):
    return _under_trial(experiment_name, trial_name, "update_weights_from_ipc", model_version)


def model_version(experiment_name, trial_name, model_name):
    return _under_trial(experiment_name, trial_name, "model_version", model_name)


def training_samples(experiment_name, trial_name):
    return _under_trial(experiment_name, trial_name, "training_samples")


# ---- workers of the worker runtime
def worker_root(experiment_name, trial_name):
    return _under_trial(experiment_name, trial_name, "worker", trailing_slash=True)


def worker(experiment_name, trial_name, worker_type, worker_index):
    return _under_trial(experiment_name, trial_name, "worker", worker_type, worker_index)


def worker_status(experiment_name, trial_name, worker_name):
    return _under_trial(experiment_name, trial_name, "status", worker_name)


def worker_key(experiment_name, trial_name, key):
    return _under_trial(experiment_name, trial_name, "worker_key", key)


def experiment_status(experiment_name, trial_name):
    return _under_trial(experiment_name, trial_name, "experiment_status")


# ---- streams and process groups
def push_pull_stream_root(experiment_name, trial_name):
    return _under_trial(experiment_name, trial_name, "push_pull_stream", trailing_slash=True)


def push_pull_stream(experiment_name, trial_name, stream_name):
    return _under_trial(experiment_name, trial_name, "push_pull_stream", stream_name)


def request_reply_stream(experiment_name, trial_name, stream_name):
    return _under_trial(experiment_name, trial_name, "request_reply_stream", stream_name)


def stream_pullers(experiment_name, trial_name):
    return _under_trial(experiment_name, trial_name, "stream_pullers")


def distributed_peer(experiment_name, trial_name, model_name):
    return _under_trial(experiment_name, trial_name, "distributed", "peer", model_name)


def distributed_master(experiment_name, trial_name, model_name):
    return _under_trial(experiment_name, trial_name, "distributed", "master", model_name)


# ---- per-host bookkeeping and metrics
def used_ports(experiment_name, trial_name, host_name):
    return _under_trial(experiment_name, trial_name, host_name, trailing_slash=True)


def metric_server(experiment_name, trial_name, group, name):
    return _under_trial(experiment_name, trial_name, "metrics", group, name)
