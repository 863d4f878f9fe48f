"""orbx_pack_records_device (the one-kernel record pack of the multi-GPU step) against the byte layout that
batching.unpack_records / the gloo CPU path define."""
import importlib
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
batching = importlib.import_module("orb_slam2v2-1_amd.batching")


@pytest.mark.parametrize("B,cap", [(1, 1), (3, 37), (64, 1260)])
def test_pack_records_device_matches_layout(B, cap):
    g = torch.Generator().manual_seed(B * 1000 + cap)
    kps = torch.randn((B, cap, 7), generator=g)
    desc = torch.randint(0, 256, (B, cap, 32), generator=g, dtype=torch.uint8)
    ur, dp = torch.randn((B, cap), generator=g), torch.randn((B, cap), generator=g)
    cnt = torch.randint(0, cap + 1, (B,), generator=g, dtype=torch.int32)
    want = batching.pack_records(kps, desc, ur, dp, cnt)                    # CPU tensors: torch slicing
    out = torch.full((B, batching.record_bytes(cap)), 0xAB, dtype=torch.uint8, device="cuda")
    got = batching.pack_records(kps.cuda(), desc.cuda(), ur.cuda(), dp.cuda(), cnt.cuda(), out=out)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(got.cpu().numpy(), want.numpy())
    u = batching.unpack_records(got, cap)
    assert torch.equal(u["counts"].cpu(), cnt) and torch.equal(u["desc"].cpu(), desc)
    assert torch.equal(u["kps"].cpu().view(torch.int32), kps.view(torch.int32))


def test_pack_records_mono_null_pointers():
    pkg = importlib.import_module("orb_slam2v2-1_amd")
    B, cap = 2, 50
    kps = torch.randn((B, cap, 7), device="cuda")
    desc = torch.randint(0, 256, (B, cap, 32), dtype=torch.uint8, device="cuda")
    cnt = torch.tensor([50, 7], dtype=torch.int32, device="cuda")
    out = torch.full((B, batching.record_bytes(cap)), 0xCD, dtype=torch.uint8, device="cuda")
    pkg.pack_records_device(kps.data_ptr(), desc.data_ptr(), 0, 0, cnt.data_ptr(), B, cap, out.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    u = batching.unpack_records(out, cap)
    assert float(u["uright"].abs().max()) == 0.0 and float(u["depth"].abs().max()) == 0.0
    assert u["counts"].tolist() == [50, 7]
    assert out[:, -12:].abs().max().item() == 0
