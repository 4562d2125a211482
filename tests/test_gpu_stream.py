"""GPU, one rank: the frame-stream host logic (x266_amd/stream.py) driving the real kernels through
the C ABI -- buffer slots, the side stream and the event ordering of the pipelined schedule."""
import numpy as np
import pytest
import torch

import x266_amd
from x266_amd.stream import FrameGeometry, PipelinedFrameStream, ShardedFrameStream

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def codec():
    return x266_amd.Codec(0)


def _fns(codec):
    stream = lambda: torch.cuda.current_stream().cuda_stream
    return (lambda tin, tout, n: codec.dct32_fwd_dev(tin.data_ptr(), tout.data_ptr(), n, stream()),
            lambda tin, tout, n: codec.satd8x8_dev(tin.data_ptr(), tout.data_ptr(), n, stream()))


@pytest.mark.parametrize("w,h", [(96, 160), (1920, 1088)])
def test_pipelined_stream_single_gpu(codec, oracle, w, h):
    dev = torch.device("cuda", 0)
    geo = FrameGeometry(w, h)
    dct_fn, satd_fn = _fns(codec)
    st = PipelinedFrameStream(geo, dev, dct_fn, satd_fn, dist=None)
    n_frames = 6
    frames = [(oracle.fill_residual(geo.dct_blocks * 1024, 0x266, f * 10 ** 7), oracle.fill_residual(geo.satd_blocks * 64, 0x267, f * 10 ** 7))
              for f in range(n_frames)]
    got = {}
    st.run(n_frames, lambda f: (torch.from_numpy(frames[f][0]).to(dev), torch.from_numpy(frames[f][1]).to(dev)),
           lambda f, c, s: got.__setitem__(f, (c.cpu().numpy(), s.cpu().numpy())))
    torch.cuda.synchronize()
    for f in range(n_frames):
        assert np.array_equal(got[f][0], oracle.dct32_fwd(frames[f][0], threads=8).ravel()), f
        assert np.array_equal(got[f][1], oracle.satd8x8(frames[f][1], threads=8).astype(np.int32)), f


def test_sharded_stream_single_gpu(codec, oracle):
    dev = torch.device("cuda", 0)
    geo = FrameGeometry(128, 64)
    dct_fn, satd_fn = _fns(codec)
    st = ShardedFrameStream(geo, dev, dct_fn, satd_fn, dist=None)
    x = oracle.fill_residual(geo.dct_blocks * 1024, 11)
    d = oracle.fill_residual(geo.satd_blocks * 64, 12)
    coef, cost = st.process(torch.from_numpy(x).to(dev), torch.from_numpy(d).to(dev))
    assert np.array_equal(coef.cpu().numpy(), oracle.dct32_fwd(x).ravel())
    assert np.array_equal(cost.cpu().numpy(), oracle.satd8x8(d).astype(np.int32))
