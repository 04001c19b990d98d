"""Harness that imports the *reference* Python modules from /root/reference (read-only mount).

Only used in THIS container to (a) validate the oracle restatements and (b) generate the golden
fixtures committed under tests/golden/.  It never travels to the GPU box and nothing in the product
imports it.  Procedure follows SURVEY.md section 8(c):
  * the mount is read-only -> PYTHONDONTWRITEBYTECODE
  * Metrics.py calls .cuda() unconditionally (Metrics.py:615-627) -> no-op shim on torch.Tensor.cuda
  * the reference loader has no map_location (Inference_QBD.py:34) -> own loader below
"""
import os
import sys

REF = os.environ.get("PMP_REFERENCE_DIR", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "Model_QBD.py"))


def load():
    """Returns (Model_QBD, Metrics, Map2Partition, Inference_QBD) reference modules."""
    import torch
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    torch.Tensor.cuda = lambda self, *a, **k: self  # CPU-only container
    import Model_QBD
    import Metrics
    import Map2Partition
    import Inference_QBD
    return Model_QBD, Metrics, Map2Partition, Inference_QBD


def load_state_dict(path):
    """torch.load with map_location + strip the DataParallel 'module.' prefix (Inference_QBD.py:28-46)."""
    import torch
    sd = torch.load(path, map_location="cpu", weights_only=False)
    if "state_dict" in sd:
        sd = sd["state_dict"]
    return {(k.split("module.", 1)[-1] if k.startswith("module.") else k): v.float().contiguous()
            for k, v in sd.items()}


def ref_net(kind, state_dict):
    """kind in {Luma_Q, Luma_MSBD, Chroma_Q, Chroma_MSBD}; returns the reference nn.Module in eval mode."""
    import torch
    M, _, _, _ = load()
    net = getattr(M, kind + "_Net")()
    net.load_state_dict({k: torch.as_tensor(v) for k, v in state_dict.items()}, strict=True)
    net.eval()
    return net
