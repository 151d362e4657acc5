"""Key schema of the name-resolve store (``areal/utils/names.py:11-28`` in the reference)."""
import getpass

USER_NAMESPACE = getpass.getuser() if hasattr(getpass, "getuser") else "user"


def registry_root(user):
    return f"trial_registry/{user}"


def trial_registry(experiment_name, trial_name):
    return f"trial_registry/{USER_NAMESPACE}/{experiment_name}/{trial_name}"


def trial_root(experiment_name, trial_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}"


def gen_servers(experiment_name, trial_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/gen_servers"


def update_weights_from_disk(experiment_name, trial_name, model_version):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/update_weights_from_disk/{model_version}"


def update_weights_from_ipc(experiment_name, trial_name, model_version):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/update_weights_from_ipc/{model_version}"


def worker_status(experiment_name, trial_name, worker_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/status/{worker_name}"


def worker_key(experiment_name, trial_name, key):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/worker_key/{key}"


def worker(experiment_name, trial_name, worker_type, worker_index):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/worker/{worker_type}/{worker_index}"


def worker_root(experiment_name, trial_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/worker/"


def experiment_status(experiment_name, trial_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/experiment_status"


def model_version(experiment_name, trial_name, model_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/model_version/{model_name}"


def training_samples(experiment_name, trial_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/training_samples"


def gen_server_manager(experiment_name, trial_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/gen_server_mgr"


def push_pull_stream(experiment_name, trial_name, stream_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/push_pull_stream/{stream_name}"


def push_pull_stream_root(experiment_name, trial_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/push_pull_stream/"


def request_reply_stream(experiment_name, trial_name, stream_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/request_reply_stream/{stream_name}"


def distributed_peer(experiment_name, trial_name, model_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/distributed/peer/{model_name}"


def distributed_master(experiment_name, trial_name, model_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/distributed/master/{model_name}"


def used_ports(experiment_name, trial_name, host_name):
    return f"{USER_NAMESPACE}/{experiment_name}/{trial_name}/{host_name}/"


def stream_pullers(experiment_name, trial_name):
    return f"{trial_root(experiment_name, trial_name)}/stream_pullers"


def metric_server(experiment_name, trial_name, group, name):
    return f"{trial_root(experiment_name, trial_name)}/metrics/{group}/{name}"
