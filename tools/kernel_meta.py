#!/usr/bin/env python3
"""Register / scratch / LDS figures of every gfx950 kernel in libmeterelf_hip.so, from the code objects' own metadata
(the amdhsa notes hipcc writes): what `llvm-readelf --notes` prints, per kernel.

    python3 tools/kernel_meta.py [path/to/lib.so] [name filter]

Used by tests/test_host_logic.py to keep scratch (a non-zero private segment, spilled registers) out of the hot-path
kernels: a kernel that spills still computes the right thing, so only a check of the build notices.
No GPU needed: the library embeds one AMDGPU ELF per translation unit in its .hip_fatbin section."""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'
EM_AMDGPU = 224
FIELDS = ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size',
          'group_segment_fixed_size', 'max_flat_workgroup_size', 'wavefront_size')


def amdgpu_elfs(blob):
    """Every ELF64 image for the AMDGPU machine embedded in `blob` (offset, bytes)."""
    out = []
    at = 0
    while True:
        at = blob.find(b'\x7fELF\x02\x01\x01', at)
        if at < 0:
            break
        hdr = blob[at:at + 64]
        if len(hdr) == 64 and struct.unpack_from('<H', hdr, 18)[0] == EM_AMDGPU:
            (shoff,) = struct.unpack_from('<Q', hdr, 40)
            (shentsize, shnum) = struct.unpack_from('<HH', hdr, 58)
            size = shoff + shentsize * shnum
            # section contents may lie behind the section header table: take the furthest extent
            for i in range(shnum):
                sh = blob[at + shoff + i * shentsize: at + shoff + (i + 1) * shentsize]
                (sh_type,) = struct.unpack_from('<I', sh, 4)
                (sh_offset, sh_size) = struct.unpack_from('<QQ', sh, 24)
                if sh_type != 8:   # SHT_NOBITS occupies no file space
                    size = max(size, sh_offset + sh_size)
            out.append((at, blob[at:at + size]))
            at += max(size, 64)
        else:
            at += 4
    return out


def kernel_metadata(lib_path=None):
    """{kernel name (demangled where the notes carry it): {field: int}} for every kernel of the library."""
    lib_path = lib_path or os.path.join(ROOT, 'meterelf_amd', 'libmeterelf_hip.so')
    with open(lib_path, 'rb') as fp:
        blob = fp.read()
    meta = {}
    for (_off, elf) in amdgpu_elfs(blob):
        with tempfile.NamedTemporaryFile(suffix='.co') as tf:
            tf.write(elf)
            tf.flush()
            txt = subprocess.run([READELF, '--notes', tf.name], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
        # the notes are a YAML document; one "- .agpr_count: ..." item per kernel under amdhsa.kernels
        for item in re.split(r'\n\s*- \.', txt):
            name = re.search(r'\.name:\s+(\S+)', item)
            if not name or '.vgpr_count' not in item:
                continue
            d = {}
            for f in FIELDS:
                m = re.search(r'\.%s:\s+(\d+)' % f, item)
                if m:
                    d[f] = int(m.group(1))
            meta[name.group(1)] = d
    return meta


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'] + list(names), stdout=subprocess.PIPE, text=True).stdout.split('\n')
        return dict(zip(names, out))
    except OSError:
        return {n: n for n in names}


if __name__ == '__main__':
    path = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else None
    flt = sys.argv[-1] if len(sys.argv) > 1 and not os.path.exists(sys.argv[-1]) else ''
    meta = kernel_metadata(path)
    pretty = demangle(sorted(meta))
    for n in sorted(meta):
        if flt and flt not in pretty[n]:
            continue
        d = meta[n]
        short = re.sub(r'\(.*', '', pretty[n]).replace('void melf::', '')
        print('%-44s vgpr %3d agpr %3d sgpr %3d  spilled v %3d s %3d  scratch %4d B  lds %6d B' % (
            short[:44], d.get('vgpr_count', 0), d.get('agpr_count', 0), d.get('sgpr_count', 0), d.get('vgpr_spill_count', 0),
            d.get('sgpr_spill_count', 0), d.get('private_segment_fixed_size', 0), d.get('group_segment_fixed_size', 0)))
