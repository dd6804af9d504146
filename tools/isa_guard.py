#!/usr/bin/env python3
"""Build-time guard for the hand-scheduled gfx950 kernels: unbundle the device code object of a hipcc object file and read
what the compiler actually produced.

    python tools/isa_guard.py sda_amd/lib/conv_wino4.o [kernel-substring]

* `kernel_metadata(obj)`   -> {mangled name: {vgpr_count, agpr_count, sgpr_count, vgpr_spill_count, sgpr_spill_count,
                                              private_segment_fixed_size, group_segment_fixed_size}}   (llvm-readelf --notes)
* `disassemble(obj)`       -> {mangled name: [instruction text, ...]}                                  (llvm-objdump -d)
* `vmem_between_waits(ins)`-> for a helper-wave instruction stream: the vector-memory loads counted between consecutive
                              `s_waitcnt vmcnt(N)` -- what the hand-written counts in conv_wino4.hip / block1d.hip rely on.

Used by tests/test_isa_guard.py (-m "not gpu"): a compiler change that introduces scratch traffic, spills, or moves a
global load across one of the hand-counted waits fails the CPU suite instead of silently corrupting / slowing the kernel.
"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = os.environ.get('SDA_LLVM_BIN', '/opt/rocm/lib/llvm/bin')
TARGET = 'hipv4-amdgcn-amd-amdhsa--gfx950'
_FIELDS = ('vgpr_count', 'agpr_count', 'sgpr_count', 'vgpr_spill_count', 'sgpr_spill_count', 'private_segment_fixed_size',
           'group_segment_fixed_size')


def _run(*cmd):
    return subprocess.run(cmd, check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE).stdout.decode(errors='replace')


def code_object(obj: str, workdir: str) -> str:
    """hipcc host object (.hip_fatbin section), offload bundle, or bare code object -> path of the gfx950 ELF."""
    head = open(obj, 'rb').read(24)
    bundle = obj
    if head.startswith(b'\x7fELF'):
        sections = _run(f'{LLVM}/llvm-readelf', '-S', obj)
        if '.hip_fatbin' not in sections:
            return obj                                   # already a device code object
        bundle = os.path.join(workdir, 'fat.bin')
        _run(f'{LLVM}/llvm-objcopy', '--dump-section', f'.hip_fatbin={bundle}', obj)
    out = os.path.join(workdir, 'dev.co')
    _run(f'{LLVM}/clang-offload-bundler', '--unbundle', '--type=o', f'--targets={TARGET}', f'--input={bundle}', f'--output={out}')
    return out


def kernel_metadata(obj: str) -> dict:
    with tempfile.TemporaryDirectory() as wd:
        notes = _run(f'{LLVM}/llvm-readelf', '--notes', code_object(obj, wd))
    kernels, cur = {}, None
    for line in notes.split('\n'):
        t = line.strip()
        if t.startswith('- .') or t.startswith('-   .'):      # a new entry of amdhsa.kernels (or of an args list)
            t = t[1:].strip()
            if cur is not None and '.name' in cur and '.vgpr_count' in cur:
                kernels[cur['.name']] = cur
            if t.startswith('.agpr_count') or cur is None:
                cur = {}
        m = re.match(r'(\.[a-z_]+):\s+(.*)$', t)
        if m and cur is not None:
            cur.setdefault(m.group(1), m.group(2).strip())
    if cur is not None and '.name' in cur and '.vgpr_count' in cur:
        kernels[cur['.name']] = cur
    return {name: {f: int(k['.' + f]) for f in _FIELDS if '.' + f in k} for name, k in kernels.items()}


def disassemble(obj: str) -> dict:
    with tempfile.TemporaryDirectory() as wd:
        text = _run(f'{LLVM}/llvm-objdump', '-d', '--no-show-raw-insn', code_object(obj, wd))
    out, cur = {}, None
    for line in text.split('\n'):
        m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        t = line.split('//')[0].strip()
        if cur is not None and t and not t.endswith(':'):
            cur.append(t)
    return out


def histogram(ins) -> dict:
    h = {}
    for i in ins:
        op = i.split()[0]
        h[op] = h.get(op, 0) + 1
    return h


def is_vmem_load(i: str) -> bool:
    op = i.split()[0]
    return op.startswith('global_load') or op.startswith('buffer_load') or op.startswith('scratch_load') or op.startswith('flat_load')


def vmem_between_waits(ins):
    """[(vmcnt N of the wait, vector-memory loads issued since the previous vmcnt wait), ...] in program order."""
    out, n = [], 0
    for i in ins:
        if is_vmem_load(i):
            n += 1
        elif i.startswith('s_waitcnt') and 'vmcnt' in i:
            m = re.search(r'vmcnt\((\d+)\)', i)
            out.append((int(m.group(1)), n))
            n = 0
    return out


def summary(obj: str, pat: str = ''):
    md = kernel_metadata(obj)
    dis = disassemble(obj)
    for name in sorted(md):
        if pat and pat not in name:
            continue
        h = histogram(dis.get(name, []))
        k = md[name]
        print(f"{name}\n   vgpr {k.get('vgpr_count')} agpr {k.get('agpr_count')} sgpr {k.get('sgpr_count')} | spills v {k.get('vgpr_spill_count')} "
              f"s {k.get('sgpr_spill_count')} | scratch {k.get('private_segment_fixed_size')} B | lds {k.get('group_segment_fixed_size')} B | "
              f"instr {sum(h.values())} mfma {sum(v for o, v in h.items() if 'mfma' in o)} scratch-ops {sum(v for o, v in h.items() if o.startswith('scratch_'))} "
              f"global_load {sum(v for o, v in h.items() if o.startswith('global_load'))} vmcnt(0) {sum(1 for i in dis.get(name, []) if i.startswith('s_waitcnt') and 'vmcnt(0)' in i)}")


if __name__ == '__main__':
    summary(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
