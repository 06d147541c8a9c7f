"""Helpers shared by the tests: golden-vector loading and small adapters."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

PARAMS = ("x_center", "x_support", "edge_attr_support", "p_support",
          "support_attr_sc_weight", "center_attr_sc_weight", "edge_attr_support_sc_weight")


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def group(flat, prefix):
    """Sub-dict of keys under ``prefix/`` with the prefix stripped, as torch tensors."""
    pre = prefix + "/"
    out = {}
    for k, v in flat.items():
        if k.startswith(pre):
            out[k[len(pre):]] = torch.from_numpy(np.asarray(v)) if v.dtype.kind in "fiub" else v
    return out


def case_names(flat):
    return sorted({k.split("/")[0] for k in flat if "/" in k})


def kc_params(case):
    return {k[len("param_"):]: v for k, v in case.items() if k.startswith("param_")}


def kc_inputs(case):
    return (case["x_focal"], case["p_focal"], case["x_neighbor"], case["p_neighbor"],
            case["edge_attr_neighbor"], bool(int(case["is_last_layer"])))


class Bag:
    def __init__(self, **kw):
        self.__dict__.update(kw)

    def keys(self):
        return list(self.__dict__)


def batch_from(flat, prefix="in_"):
    return Bag(**{k[len(prefix):]: torch.from_numpy(np.asarray(v)) for k, v in flat.items() if k.startswith(prefix)})
