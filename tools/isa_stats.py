#!/usr/bin/env python
"""Instruction-mix summary of the gfx950 code object inside a HIP shared library (CPU-only: no GPU needed).

    python tools/isa_stats.py contact-human-dynamics_amd/csrc/libchd_phys.so [--functions]

Prints, for the whole object and (with --functions) per function: flat / global / scratch / LDS memory instructions,
fp64 MFMA instructions, and from the kernel metadata the private (scratch) segment size, VGPR / SGPR counts and LDS size.
"""
import collections
import re
import subprocess
import sys
import tempfile
import os

LLVM = '/opt/rocm/lib/llvm/bin'


def extract(so, tmp):
    out = os.path.join(tmp, 'dev.co')
    subprocess.check_call([os.path.join(LLVM, 'clang-offload-bundler'), '--unbundle', '--type=o', '--input=' + so,
                           '--targets=hipv4-amdgcn-amd-amdhsa--gfx950', '--output=' + out], stderr=subprocess.DEVNULL)
    return out


def main():
    so = sys.argv[1]
    per_fn = '--functions' in sys.argv
    with tempfile.TemporaryDirectory() as tmp:
        # the fat binary sits in the .hip_fatbin section of the shared library
        fat = os.path.join(tmp, 'fat.bin')
        subprocess.check_call([os.path.join(LLVM, 'llvm-objcopy'), '-O', 'binary', '--only-section=.hip_fatbin', so, fat])
        co = extract(fat, tmp)
        dis = subprocess.run([os.path.join(LLVM, 'llvm-objdump'), '-d', '--no-show-raw-insn', co], stdout=subprocess.PIPE, text=True).stdout
        notes = subprocess.run([os.path.join(LLVM, 'llvm-readelf'), '--notes', co], stdout=subprocess.PIPE, text=True).stdout
    kinds = [('flat', r'\bflat_(load|store|atomic)'), ('global', r'\bglobal_(load|store|atomic)'), ('scratch', r'\bscratch_(load|store)'),
             ('buffer', r'\bbuffer_(load|store)'), ('lds', r'\bds_(read|write|load|store|bpermute|swizzle)'), ('mfma_f64', r'v_mfma_f64'), ('s_load', r'\bs_load_')]
    tot = collections.Counter(); fn = None; per = collections.defaultdict(collections.Counter)
    for line in dis.splitlines():
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            fn = m.group(1); continue
        for k, pat in kinds:
            if re.search(pat, line):
                tot[k] += 1
                if fn:
                    per[fn][k] += 1
        tot['insns'] += 1
        if fn:
            per[fn]['insns'] += 1
    print('whole object: ' + '  '.join('%s %d' % (k, tot[k]) for k, _ in kinds) + '  instructions %d' % tot['insns'])
    for blk in re.split(r'\n\s*- ', notes):
        name = re.search(r'\.name:\s+(\S+)', blk)
        if not name or '.private_segment_fixed_size' not in blk:
            continue
        g = lambda key: (re.search(r'\.%s:\s+(\d+)' % key, blk) or [None, '?'])[1]       # noqa: E731
        print('kernel %s: private_segment %s B/lane  vgpr %s agpr %s sgpr %s  static LDS %s B  spills vgpr %s sgpr %s' % (
            name.group(1), g('private_segment_fixed_size'), g('vgpr_count'), g('agpr_count'), g('sgpr_count'), g('group_segment_fixed_size'),
            g('vgpr_spill_count'), g('sgpr_spill_count')))
    if per_fn:
        for f, c in sorted(per.items(), key=lambda kv: -kv[1]['insns']):
            if c['insns'] < 200:
                continue
            print('%-90s insns %6d flat %5d global %5d scratch %5d lds %5d mfma %3d' % (f[:90], c['insns'], c['flat'], c['global'], c['scratch'], c['lds'], c['mfma_f64']))


if __name__ == '__main__':
    main()
