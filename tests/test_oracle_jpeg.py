"""JPEG decode stage, CPU side: (1) the oracle (oracle/jpeg_oracle.py: libjpeg's default path restated in numpy) is pinned against
the pixels Pillow itself decoded from the committed fixtures (tests/golden/jpeg_cases.npz) - bit-exact; (2) the host half of the
product decoder (csrc/jpeg.hip: marker parsing + Huffman stage, plain C++ inside libofb_hip.so, no GPU involved) hands over exactly
the oracle's coefficients; (3) files outside the decoder's scope are rejected, not mis-decoded."""
import ctypes as C
import os

import numpy as np
import pytest

from oracle import jpeg_oracle as J

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'jpeg_cases.npz')


def _cases():
    z = np.load(GOLDEN)
    return [(k[:-4], z[k].tobytes(), z[k[:-4] + '.rgb']) for k in z.files if k.endswith('.jpg') and k != 'progressive.jpg' and not k.startswith('oos_')]


@pytest.mark.parametrize('name,data,rgb', _cases(), ids=[c[0] for c in _cases()])
def test_oracle_matches_pillow_fixture(name, data, rgb):
    got = J.decode(data)
    assert got.shape == rgb.shape and got.dtype == np.uint8
    assert np.array_equal(got, rgb), f'{name}: {int((got != rgb).sum())} bytes differ from Pillow\'s decode'


def test_oracle_matches_the_installed_pillow():
    """the fixtures were made by the Pillow of the build container; whatever Pillow runs the tests must agree too (<= 1 LSB:
    libjpeg-turbo releases keep JDCT_ISLOW / fancy upsampling bit-stable, the allowance covers a differently built decoder)"""
    Image = pytest.importorskip('PIL.Image')
    import io
    for name, data, _ in _cases():
        ref = np.asarray(Image.open(io.BytesIO(data)).convert('RGB')).astype(int)
        assert np.abs(J.decode(data).astype(int) - ref).max() <= 1, name


@pytest.mark.parametrize('name,data,rgb', _cases(), ids=[c[0] for c in _cases()])
def test_host_entropy_stage_matches_oracle(name, data, rgb):
    import __graft_entry__ as g
    g.build()
    from ofb_amd import hip
    info, buf = hip.jpeg_parse(data)
    f = J.parse_and_decode(data)
    assert (info.width, info.height, info.ncomp) == (f['width'], f['height'], len(f['comps']))
    coef = np.zeros(int(info.coef_count), np.int16)
    hip.jpeg_decode_coefficients(buf, len(data), info, coef.ctypes.data)
    for c, comp in enumerate(f['comps']):
        assert (info.hs[c], info.vs[c], info.blocks_w[c], info.blocks_h[c]) == (comp['h'], comp['v'], comp['bw'], comp['bh'])
        got = coef[info.coef_off[c]:info.coef_off[c] + comp['bw'] * comp['bh'] * 64].reshape(comp['bh'], comp['bw'], 64)
        assert np.array_equal(got.astype(np.int64), comp['coef']), (name, c)
        assert np.array_equal(np.array(info.quant[c][:], np.int64), comp['quant']), (name, c)


def test_out_of_scope_files_are_rejected():
    import __graft_entry__ as g
    g.build()
    from ofb_amd import hip
    z = np.load(GOLDEN)
    with pytest.raises(hip.OfbError):
        hip.jpeg_parse(z['progressive.jpg'].tobytes())          # SOF2: OFB_ELIMIT
    for k in ('oos_progressive.jpg', 'oos_cmyk.jpg', 'oos_adobe_rgb.jpg'):   # SOF2; four components; Adobe transform 0 = RGB planes
        with pytest.raises(hip.OfbError):
            hip.jpeg_parse(z[k].tobytes())
    with pytest.raises(hip.OfbError):
        hip.jpeg_parse(b'\x89PNG\r\n\x1a\n' + bytes(32))         # not a JPEG
    good = z['q75_420_53x37.jpg'].tobytes()
    info, buf = hip.jpeg_parse(good)
    bad = bytearray(good)
    bad[len(bad) // 2:] = bytes(len(bad) - len(bad) // 2)         # truncated / zeroed entropy data: must fail or finish, never crash
    info2, buf2 = hip.jpeg_parse(bytes(bad))
    coef = np.zeros(int(info2.coef_count), np.int16)
    try:
        hip.jpeg_decode_coefficients(buf2, len(bad), info2, coef.ctypes.data)
    except hip.OfbError:
        pass


def test_batch_host_stage_equals_the_per_file_calls():
    """ofb_jpeg_plan_batch / ofb_jpeg_decode_batch (native thread pool, what JpegDecoder uses) against the per-file entry points"""
    import __graft_entry__ as g
    g.build()
    from ofb_amd import hip
    cases = _cases()
    blobs = [c[1] for c in cases] * 3                                  # 36 files over 5 threads: the work queue wraps around
    pb = hip.jpeg_plan_batch(blobs)
    coef = np.full(pb.coef_total, 77, np.int16)
    hip.jpeg_decode_batch(pb, coef.ctypes.data, 5)
    out_off, plane_off = 0, 0
    for i, data in enumerate(blobs):
        info, buf = hip.jpeg_parse(data)
        one = np.zeros(int(info.coef_count), np.int16)
        hip.jpeg_decode_coefficients(buf, len(data), info, one.ctypes.data)
        j = pb.jobs[i]
        assert (j.width, j.height, j.ncomp, j.out_off) == (info.width, info.height, info.ncomp, out_off)
        base = j.coef_off[0] - info.coef_off[0]
        assert np.array_equal(coef[base:base + int(info.coef_count)], one), i
        for c in range(info.ncomp):
            assert j.coef_off[c] == base + info.coef_off[c] and j.plane_off[c] == plane_off
            assert list(j.quant[c]) == list(info.quant[c])
            plane_off += (info.blocks_w[c] * info.blocks_h[c] * 64 + 15) // 16 * 16
        out_off += (info.height * info.width * 3 + 15) // 16 * 16
    assert (pb.out_total, pb.plane_total) == (out_off, plane_off)
    z = np.load(GOLDEN)
    with pytest.raises(hip.OfbError):
        hip.jpeg_plan_batch([blobs[0], z['progressive.jpg'].tobytes()])
