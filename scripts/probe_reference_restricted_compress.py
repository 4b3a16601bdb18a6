"""Build-container probe (needs /root/reference): what does the REFERENCE's MAESparseAttention.compress (models/layers.py:559-690) do
in the head-only (--head_search) and channel-only (--channel_search) attention spaces?  once-for-both_amd raises NotImplementedError
there (layers.py: compress of a restricted attention space); this script shows there is nothing to be compatible with:
  head-only:    compress cuts qkv to 3 * heads * ONE channel (channel_index comes from the (H, 1) score) -> the next forward raises
  channel-only: compress itself raises (index 1 is out of bounds: the (1, n) alpha is indexed with the head index list)
Run:  python scripts/probe_reference_restricted_compress.py       (recorded output: profiles/r05_reference_restricted_compress_probe.txt)"""
import contextlib
import io
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import make_golden as G                                    # noqa: E402  (imports the reference)
from oracle import ofb_oracle as O                         # noqa: E402

for space in ('head', 'channel'):
    cfg = O.Config(**dict(embed_dim=64, depth=2, num_heads=4, num_classes=10, attn_space=space), drop_path_rate=0.0)
    model = G.build_reference(cfg, 0.0)
    a = model.blocks[0].attn
    print(space, 'alpha', tuple(a.alpha.shape), 'score', tuple(a.score.shape), 'mask', tuple(a.mask.shape))
    with torch.no_grad():
        a.alpha.fill_(0.2)
        a.alpha.view(-1)[-1] = -6.0                         # the last cell dies -> the "last row / column off" cut
    torch.cuda.synchronize = lambda *x, **k: None
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            a.compress(0.2, None, None, None, prefix='blocks.0.attn')
        print('  after compress: qkv.weight', tuple(a.qkv.weight.shape), 'proj.weight', tuple(a.proj.weight.shape), 'score', tuple(a.score.shape),
              'head_num', getattr(a, 'head_num', None), 'num_heads', a.num_heads, 'head_dim', a.head_dim)
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                y = a(torch.randn(2, 197, 64))
            print('  forward after the cut: ok', tuple(y.shape))
        except Exception as e:                             # noqa: BLE001
            print('  forward after the cut FAILS in the reference:', type(e).__name__, str(e)[:200])
    except Exception as e:                                 # noqa: BLE001
        print('  compress FAILS in the reference:', type(e).__name__, str(e)[:300])
