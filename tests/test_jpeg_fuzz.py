"""The JPEG host stage (csrc/jpeg_host.cpp: marker parsing + Huffman decoding) reads UNTRUSTED bytes - ImageNet files as they come
(reference datasets.py:90-125 hands the same bytes to libjpeg).  The very source file the library ships is built here with
g++ -fsanitize=address,undefined (sanitizers run on the CPU build only) and fed truncated, bit-flipped, byte-shuffled and
table-corrupted variants of the committed fixtures: every call must return (OK or an error code) without a sanitizer report."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden', 'jpeg_cases.npz')

DRIVER = r'''
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "include/ofb_hip.h"
// file format: u32 count, then per case u32 nbytes + bytes
int main(int argc, char** argv) {
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 2;
  uint32_t n = 0;
  if (fread(&n, 4, 1, f) != 1) return 2;
  long ok = 0, rejected = 0;
  for (uint32_t i = 0; i < n; ++i) {
    uint32_t len = 0;
    if (fread(&len, 4, 1, f) != 1) return 2;
    // an exactly-sized heap block: any read past the file is an ASan report
    uint8_t* buf = (uint8_t*)malloc(len ? len : 1);
    if (len && fread(buf, 1, len, f) != len) return 2;
    ofb_jpeg_info info;
    int rc = ofb_jpeg_parse(buf, len, &info);
    if (rc == 0 && info.coef_count > 0 && info.coef_count < (int64_t)1 << 26) {
      std::vector<int16_t> coef((size_t)info.coef_count);
      rc = ofb_jpeg_decode_coefficients(buf, len, &info, coef.data());
    }
    if (rc == 0) ++ok; else ++rejected;
    free(buf);
  }
  printf("cases %u ok %ld rejected %ld\n", n, ok, rejected);
  return 0;
}
'''


def _variants(rng, data, n):
    out = []
    b = np.frombuffer(data, np.uint8)
    for i in range(n):
        v = b.copy()
        kind = i % 6
        if kind == 0:                                          # truncation anywhere
            v = v[:rng.integers(0, len(v))]
        elif kind == 1:                                        # random bit flips
            for _ in range(rng.integers(1, 8)):
                v[rng.integers(0, len(v))] ^= 1 << rng.integers(0, 8)
        elif kind == 2:                                        # random bytes in the header region (tables, frame / scan headers)
            for _ in range(rng.integers(1, 6)):
                v[rng.integers(2, min(len(v), 700))] = rng.integers(0, 256)
        elif kind == 3:                                        # a run of 0xFF (marker storms) / zeros inside the entropy data
            p = rng.integers(len(v) // 2, len(v))
            v[p:p + rng.integers(1, 40)] = 0xFF if rng.integers(0, 2) else 0
        elif kind == 4:                                        # shuffled tail
            p = rng.integers(2, len(v))
            rng.shuffle(v[p:])
        else:                                                  # segment lengths overwritten
            idx = np.flatnonzero(v[:-3] == 0xFF)
            if len(idx):
                p = idx[rng.integers(0, len(idx))]
                v[p + 2:p + 4] = rng.integers(0, 256, 2)
        out.append(v.tobytes())
    return out


@pytest.mark.skipif(shutil.which('g++') is None, reason='needs g++')
def test_host_stage_survives_corrupted_files_under_sanitizers(tmp_path):
    src = os.path.join(ROOT, 'once-for-both_amd', 'csrc', 'jpeg_host.cpp')
    drv = tmp_path / 'fuzz_driver.cpp'
    drv.write_text(DRIVER)
    exe = tmp_path / 'jpeg_fuzz'
    cmd = ['g++', '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=all', '-fno-omit-frame-pointer',
           '-I', ROOT, str(drv), src, '-o', str(exe), '-lpthread']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(GOLDEN)
    rng = np.random.default_rng(2024)
    cases = []
    for k in z.files:
        if k.endswith('.jpg'):
            data = z[k].tobytes()
            cases.append(data)                                 # the intact file first
            cases += _variants(rng, data, 160)
    cases += [b'', b'\xff', b'\xff\xd8', b'\xff\xd8\xff', b'\xff\xd8\xff\xda\x00\x02', bytes(64), b'\xff' * 300]
    blob = tmp_path / 'cases.bin'
    with open(blob, 'wb') as f:
        f.write(np.uint32(len(cases)).tobytes())
        for c in cases:
            f.write(np.uint32(len(c)).tobytes())
            f.write(c)
    env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0', UBSAN_OPTIONS='print_stacktrace=1')
    r = subprocess.run([str(exe), str(blob)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-4000:])
    print(r.stdout.strip())
    n_ok = int(r.stdout.split('ok')[1].split()[0])
    assert n_ok >= 12, 'the intact fixtures must still decode'
