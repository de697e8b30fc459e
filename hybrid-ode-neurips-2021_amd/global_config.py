"""Global dtype / device choice (mirrors the role of reference ``global_config.py:3-8``).

The reference hard-codes ``cuda:1``; here the device follows the launcher: one process per GPU, ``LOCAL_RANK`` picks
the card (``HODE_DEVICE`` overrides), CPU when no HIP device is visible (host-logic tests only -- the solver path
itself refuses CPU tensors).
"""
import os

import torch

DTYPE = torch.float32


def get_device():
    forced = os.environ.get("HODE_DEVICE")
    if forced:
        return torch.device(forced)
    if torch.cuda.is_available():
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    return torch.device("cpu")
