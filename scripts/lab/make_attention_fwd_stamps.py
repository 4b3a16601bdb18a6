"""Lab: writes scripts/lab/bin/attention_fwd_stamps.hip = the product's attention.hip with s_memtime stamps in the FORWARD kernel (one mid-grid
workgroup, every wave: start, after the prologue barrier, five per key block, loop end, after the f32 stores, end), builds it against the tree's
other objects into scripts/lab/bin/libofb_attfstamps.so.  Read the stamps with scripts/lab/stamp_att_fwd.py.
usage (build container): python scripts/lab/make_attention_fwd_stamps.py"""
import os, subprocess, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
C = os.path.join(R, 'once-for-both_amd', 'csrc')
s = open(os.path.join(C, 'attention.hip')).read()

def sub(old, new):
    global s
    assert old in s, old[:60]
    s = s.replace(old, new, 1)

infra = '''
__device__ unsigned long long ofb_attf_stamps[13 * 40];
#define AF_STAMP(slot) do { if (blockIdx.x == gridDim.x / 2 + 3 && (threadIdx.x & 63) == 0 && (slot) < 40) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ofb_attf_stamps[(threadIdx.x >> 6) * 40 + (slot)] = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" int ofb_diag_attf_stamps(unsigned long long* out_host) { return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(ofb_attf_stamps), sizeof(unsigned long long) * 13 * 40); }
'''
anchor = '// ------------------------------------------------------------------------------------------------------------------\n// forward kernel'
sub(anchor, infra + anchor)
sub('  const int t = threadIdx.x, lane = t & 63, w = t >> 6, c = lane & 15, g = lane >> 4;\n  const int b = blockIdx.x / H, head = blockIdx.x % H;\n  const int wq = blockIdx.y * ATT_NT + w;',
    '  AF_STAMP(0);\n  const int t = threadIdx.x, lane = t & 63, w = t >> 6, c = lane & 15, g = lane >> 4;\n  const int b = blockIdx.x / H, head = blockIdx.x % H;\n  const int wq = blockIdx.y * ATT_NT + w;')
sub('  __syncthreads();\n\n  const bool active = wq * ATT_T < N + sft;', '  __syncthreads();\n  AF_STAMP(1);\n\n  const bool active = wq * ATT_T < N + sft;')
sub('      // online softmax: this lane holds keys kb*32 + 16 tk + 4g + r of query c', '      AF_STAMP(2 + 5 * kb);\n      // online softmax: this lane holds keys kb*32 + 16 tk + 4g + r of query c')
sub('      att_hx8 pf[2];\n      att_split8(p8, pf);', '      att_hx8 pf[2];\n      att_split8(p8, pf);\n      AF_STAMP(3 + 5 * kb);')
sub('    if (kb + 1 < nb) stage_store((kb + 1) & 1);\n    __syncthreads();\n  }\n  if (!active) return;',
    '    AF_STAMP(4 + 5 * kb);\n    if (kb + 1 < nb) stage_store((kb + 1) & 1);\n    AF_STAMP(5 + 5 * kb);\n    __syncthreads();\n    AF_STAMP(6 + 5 * kb);\n  }\n  AF_STAMP(37);\n  if (!active) return;')
sub('  if (PF) {\n    // wave-private patch [32 ch][16 q + 4] f32 in the (now free) stage buffers: 2.5 KB per wave',
    '  AF_STAMP(38);\n  if (PF) {\n    // wave-private patch [32 ch][16 q + 4] f32 in the (now free) stage buffers: 2.5 KB per wave')
sub('      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");\n      __builtin_amdgcn_wave_barrier();\n    }\n  }\n}\n\n// ------------------------------------------------------------------------------------------------------------------\n// backward',
    '      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");\n      __builtin_amdgcn_wave_barrier();\n    }\n  }\n  AF_STAMP(39);\n}\n\n// ------------------------------------------------------------------------------------------------------------------\n// backward')
out = os.path.join(R, 'scripts', 'lab', 'bin')
os.makedirs(out, exist_ok=True)
open(os.path.join(out, 'attention_fwd_stamps.hip'), 'w').write(s)
objs = [o for o in glob.glob(os.path.join(C, 'build', '*.o')) if not o.endswith('attention.o')]
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-function', '-I', C, '-c',
                       os.path.join(out, 'attention_fwd_stamps.hip'), '-o', os.path.join(out, 'attention_fwd_stamps.o')])
subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', os.path.join(out, 'libofb_attfstamps.so'),
                       os.path.join(out, 'attention_fwd_stamps.o')] + objs)
print('built', os.path.join(out, 'libofb_attfstamps.so'))
