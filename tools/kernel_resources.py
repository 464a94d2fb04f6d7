#!/usr/bin/env python3
"""Static resource audit of every gfx950 kernel inside a built HIP shared library -- no GPU, no external tool.

Why it exists ("a perf regression sits on the final tree"): a source-level refactor of the thread-fused
pointwise body made LLVM keep 24 / 48 bytes of a per-thread array in memory; AMDGPUPromoteAlloca moved them into LDS
(3-channel kernels: +6 KB per workgroup and a dispatch-packet read for the linear thread id) or into scratch (4-channel
kernels).  Every launch of those kernels became 20-25 us slower and 2435 bit-exactness tests stayed green.  All three
symptoms are visible in the code object's metadata, so the CPU test suite (tests/test_kernel_resources.py) now reads it:

  * .private_segment_fixed_size > 0 or .uses_dynamic_stack   -> scratch (VGPR spills / unpromoted allocas / calls); SGPRs
    parked in VGPR lanes (.sgpr_spill_count) stay in registers and are reported, not refused
  * hidden_* / dispatch-ptr user SGPR                        -> a promoted alloca (nothing in this engine asks for the packet)
  * .group_segment_fixed_size, .vgpr_count, .sgpr_count      -> compared with the committed table (tools/kernel_resources.json)

Parsing: the .hip_fatbin section holds clang offload bundles ("__CLANG_OFFLOAD_BUNDLE__": n entries of {offset, size,
triple}); each hipv4-amdgcn entry is an ELF whose NT_AMDGPU_METADATA note (type 32, owner "AMDGPU") is a msgpack map
with 'amdhsa.kernels'; the kernel descriptor (<name>.kd in .rodata, 64 bytes) carries the user-SGPR enable bits.

usage: kernel_resources.py [lib.so] [--json out.json] [--check table.json]
"""
import json
import os
import struct
import subprocess
import sys

import msgpack

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_LIB = os.path.join(ROOT, "cvgpuspeedup_amd", "lib", "libcvgs_hip.so")
TABLE = os.path.join(ROOT, "tools", "kernel_resources.json")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _elf_sections(buf):
    """[(name, type, offset, size, addr)] of a 64-bit little-endian ELF held in `buf`."""
    assert buf[:4] == b"\x7fELF" and buf[4] == 2, "not an ELF64"
    shoff, = struct.unpack_from("<Q", buf, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", buf, 0x3A)
    raw = []
    for i in range(shnum):
        name, typ, _flags, addr, off, size = struct.unpack_from("<IIQQQQ", buf, shoff + i * shentsize)
        raw.append((name, typ, off, size, addr))
    stroff = raw[shstrndx][2]
    out = []
    for name, typ, off, size, addr in raw:
        end = buf.index(b"\0", stroff + name)
        out.append((buf[stroff + name:end].decode(), typ, off, size, addr))
    return out


def _bundles(fatbin):
    """Yield (triple, bytes) for every entry of every offload bundle in the .hip_fatbin section."""
    pos = 0
    while True:
        pos = fatbin.find(MAGIC, pos)
        if pos < 0:
            return
        n, = struct.unpack_from("<Q", fatbin, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", fatbin, p)
            triple = fatbin[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            yield triple, fatbin[pos + off:pos + off + size]
        pos += len(MAGIC)


def _notes(buf, off, size):
    p, end = off, off + size
    while p + 12 <= end:
        namesz, descsz, typ = struct.unpack_from("<III", buf, p)
        p += 12
        name = buf[p:p + namesz].rstrip(b"\0")
        p += (namesz + 3) & ~3
        desc = buf[p:p + descsz]
        p += (descsz + 3) & ~3
        yield name, typ, desc


def _symbols(buf, sections):
    """name -> (value, size) from .symtab / .dynsym."""
    out = {}
    for name, typ, off, size, _addr in sections:
        if typ not in (2, 11):
            continue
        strsec = ".strtab" if typ == 2 else ".dynstr"
        stroff = next(s[2] for s in sections if s[0] == strsec)
        for i in range(size // 24):
            st_name, _info, _other, _shndx, value, ssize = struct.unpack_from("<IBBHQQ", buf, off + i * 24)
            end = buf.index(b"\0", stroff + st_name)
            out[buf[stroff + st_name:end].decode()] = (value, ssize)
    return out


def kernels_of(lib_path):
    """One dict per gfx950 kernel in the library: name, lds, scratch, vgpr, sgpr, dynamic_stack, dispatch_ptr, queue_ptr, kernarg."""
    data = open(lib_path, "rb").read()
    fat = next((s for s in _elf_sections(data) if s[0] == ".hip_fatbin"), None)
    if fat is None:
        raise RuntimeError("%s has no .hip_fatbin section" % lib_path)
    fatbin = data[fat[2]:fat[2] + fat[3]]
    out = []
    for triple, co in _bundles(fatbin):
        if "amdgcn" not in triple or not co.startswith(b"\x7fELF"):
            continue
        if "gfx950" not in triple:
            raise RuntimeError("unexpected device target in %s: %s (the engine is gfx950 only)" % (lib_path, triple))
        secs = _elf_sections(co)
        syms = _symbols(co, secs)
        # virtual address -> file offset (descriptors live in .rodata)
        def file_off(va):
            for _n, _t, off, size, addr in secs:
                if addr and addr <= va < addr + size:
                    return off + (va - addr)
            raise KeyError(va)
        meta = None
        for name, typ, off, size, _addr in secs:
            if typ != 7:  # SHT_NOTE
                continue
            for owner, ntype, desc in _notes(co, off, size):
                if owner == b"AMDGPU" and ntype == 32:
                    meta = msgpack.unpackb(desc, raw=False, strict_map_key=False)
        if meta is None:
            continue
        for k in meta.get("amdhsa.kernels", []):
            kd_va, _ = syms[k[".symbol"]]
            kd = co[file_off(kd_va):file_off(kd_va) + 64]
            # kernel descriptor: compute_pgm_rsrc2 at byte 52, kernel_code_properties (u16) at byte 56
            props, = struct.unpack_from("<H", kd, 56)
            preload, = struct.unpack_from("<H", kd, 58)  # kernarg_preload: length in dwords (bits 0-6), offset (bits 7-15)
            out.append({
                "name": k[".name"],
                "lds": k[".group_segment_fixed_size"],
                "scratch": k[".private_segment_fixed_size"],
                "vgpr": k[".vgpr_count"],
                "sgpr": k[".sgpr_count"],
                "spill": k.get(".vgpr_spill_count", 0),            # VGPRs spilled to scratch MEMORY
                "sgpr_spill": k.get(".sgpr_spill_count", 0),        # SGPRs parked in VGPR lanes (v_writelane): registers, not memory
                "dynamic_stack": bool(k.get(".uses_dynamic_stack", False)),
                "dispatch_ptr": bool(props & 0x2),   # ENABLE_SGPR_DISPATCH_PTR
                "queue_ptr": bool(props & 0x4),      # ENABLE_SGPR_QUEUE_PTR
                "kernarg": k[".kernarg_segment_size"],
                "preload": preload & 0x7f,           # leading kernel-argument dwords delivered in user SGPRs with the dispatch
            })
    return out


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
        return p.stdout.split("\n")[:len(names)]
    except Exception:
        return list(names)


def violations(kernels):
    """Hard rules: no kernel of this engine may use scratch, a dynamic stack, or the dispatch / queue packet."""
    bad = []
    for k in kernels:
        why = []
        if k["scratch"] > 0:
            why.append("scratch %d B/lane" % k["scratch"])
        if k["spill"] > 0:
            why.append("%d VGPRs spilled to scratch" % k["spill"])
        if k["dynamic_stack"]:
            why.append("dynamic stack (a call that was not inlined)")
        if k["dispatch_ptr"] or k["queue_ptr"]:
            why.append("reads the dispatch/queue packet (an alloca promoted to LDS)")
        if why:
            bad.append((k["name"], ", ".join(why)))
    return bad


def table_of(kernels):
    return {k["name"]: [k["lds"], k["vgpr"], k["sgpr"]] for k in kernels}


def drift(kernels, table, vgpr_slack=4):
    """Kernels whose LDS footprint changed at all or whose VGPR count grew by more than `vgpr_slack` against the committed
    table (new kernels are reported separately: adding one must be a deliberate, reviewed change of the table)."""
    changed, new = [], []
    for k in kernels:
        ref = table.get(k["name"])
        if ref is None:
            new.append(k["name"])
        elif k["lds"] != ref[0] or k["vgpr"] > ref[1] + vgpr_slack:
            changed.append((k["name"], ref, [k["lds"], k["vgpr"], k["sgpr"]]))
    return changed, new


def main(argv):
    lib = DEFAULT_LIB
    out_json = check = None
    i = 1
    while i < len(argv):
        if argv[i] == "--json":
            out_json = argv[i + 1]; i += 2
        elif argv[i] == "--check":
            check = argv[i + 1]; i += 2
        else:
            lib = argv[i]; i += 1
    ks = kernels_of(lib)
    bad = violations(ks)
    print("%s: %d kernels, %d with scratch / dynamic stack / packet reads" % (os.path.basename(lib), len(ks), len(bad)))
    for name, why in zip(demangle([b[0] for b in bad]), [b[1] for b in bad]):
        print("  VIOLATION %s: %s" % (name[:160], why))
    if out_json:
        with open(out_json, "w") as f:
            json.dump(table_of(ks), f, indent=0, sort_keys=True)
        print("wrote", out_json)
    rc = 1 if bad else 0
    if check:
        changed, new = drift(ks, json.load(open(check)))
        for name, ref, now in changed:
            print("  DRIFT %s: [lds, vgpr, sgpr] %s -> %s" % (demangle([name])[0][:140], ref, now))
        for name in new:
            print("  NEW (not in the table) %s" % demangle([name])[0][:160])
        rc = rc or (1 if changed or new else 0)
    return rc


if __name__ == "__main__":
    sys.exit(main(sys.argv))
