"""Valid ciphertext tensors as torch CUDA tensors, made by the product path (the helper the GPU tests and the tools under
tools/ share; bench.py itself keeps to the library's own allocator and has its own copy on DBuf)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import exp_records, form_record, hx  # noqa: E402


def encrypt_tensor_gpu(eng, torch, prm, plaintexts, r, dev):
    """c1 = h^r (shared), c2_i = f^{m_i} o pk^r (cpu_cryptosystem_tensor_ops.inl:7-15).  Returns a device int32 tensor
    of 2E records."""
    import numpy as np
    E = len(plaintexts)
    f = form_record(hx(prm["f"]["a"]), hx(prm["f"]["b"]), hx(prm["f"]["c"]))
    h = form_record(hx(prm["h"]["a"]), hx(prm["h"]["b"]), hx(prm["h"]["c"]))
    pk = form_record(hx(prm["pk"]["a"]), hx(prm["pk"]["b"]), hx(prm["pk"]["c"]))
    base = torch.from_numpy(np.concatenate([h, pk]).view(np.int32)).to(dev)
    ex = torch.from_numpy(exp_records([r]).view(np.int32)).to(dev)
    hp = torch.empty_like(base)
    eng.pow_records(base.data_ptr(), ex.data_ptr(), hp.data_ptr(), 1)
    torch.cuda.synchronize()
    em = torch.from_numpy(exp_records(plaintexts).view(np.int32)).to(dev)
    out = torch.empty(E * 2 * 168, dtype=torch.int32, device=dev)
    eng.encrypt_records(em.data_ptr(), hp.data_ptr(), f, out.data_ptr(), E, prm["k"])
    torch.cuda.synchronize()
    return out
