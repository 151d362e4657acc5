"""Run several interfaces on the same input concurrently and merge their outputs (e.g. the reference
policy's forward pass on the GPU while the rule-based reward runs on host threads). Behaviour of
``realhf/impl/model/interface/fused_interface.py`` (71 lines)."""
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Optional

from ...system.data_api import SequenceSample
from ...system.dfg import ModelInterfaceAbstraction
from ...system.model_api import Model, ModelInterface, make_interface, register_interface


class FusedThreadingForwardInterface(ModelInterface):
    def __init__(self, interfaces: Dict[str, ModelInterfaceAbstraction]):
        self.interfaces = {k: (make_interface(v) if isinstance(v, ModelInterfaceAbstraction)
                               else make_interface(ModelInterfaceAbstraction(**v))) for k, v in interfaces.items()}

    def inference(self, model: Model, data: SequenceSample, mb_spec) -> Optional[SequenceSample]:
        with ThreadPoolExecutor(max_workers=len(self.interfaces)) as ex:
            futs = {k: ex.submit(itf.inference, model, data, mb_spec) for k, itf in self.interfaces.items()}
            outs = [f.result() for f in futs.values()]
        outs = [o for o in outs if o is not None]
        if not outs:
            return None
        final = outs[0]
        for o in outs[1:]:
            final.update_(o)
        return final


register_interface("fused-threading", FusedThreadingForwardInterface)
