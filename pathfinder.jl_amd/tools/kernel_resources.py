"""Register / scratch / LDS usage of every gfx950 kernel in libpfmi.so, read from the code objects' AMDGPU metadata notes.

    python tools/kernel_resources.py [pattern]      -> table of the kernels whose demangled name contains `pattern`

libpfmi.so carries one clang offload bundle per translation unit (`__CLANG_OFFLOAD_BUNDLE__` + entry table); the gfx950 entries are ELF
code objects whose NT_AMDGPU_METADATA note (`llvm-readelf --notes`) lists, per kernel, `.vgpr_count`, `.agpr_count`, `.sgpr_count`,
`.vgpr_spill_count`, `.sgpr_spill_count`, `.private_segment_fixed_size` (scratch bytes per work-item) and `.group_segment_fixed_size`
(static LDS).  tests/test_kernel_resources.py pins these numbers for the hot kernels, so a toolchain bump that re-introduces the
spills round 4 removed fails the CPU suite instead of silently costing milliseconds.
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(os.path.dirname(HERE), "lib", "libpfmi.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
_KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
         "group_segment_fixed_size", "max_flat_workgroup_size")


def code_objects(lib=LIB):
    """the gfx950 code objects embedded in `lib` (bytes)"""
    d = open(lib, "rb").read()
    out = []
    for m in re.finditer(MAGIC, d):
        b = m.start()
        p = b + len(MAGIC)
        (ne,) = struct.unpack_from("<Q", d, p)
        p += 8
        for _ in range(ne):
            off, size, tl = struct.unpack_from("<QQQ", d, p)
            p += 24
            trip = d[p:p + tl].decode()
            p += tl
            if "gfx950" in trip and size > 0:
                out.append(d[b + off:b + off + size])
    return out


def _demangle(names):
    for tool in (os.path.join(LLVM, "llvm-cxxfilt"), "c++filt"):
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True, check=True)
            return r.stdout.split("\n")[:len(names)]
        except Exception:
            continue
    return list(names)


def kernel_resources(lib=LIB):
    """{demangled kernel name: {vgpr_count, agpr_count, ..., group_segment_fixed_size}} for every kernel of the library"""
    res = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        # the metadata is YAML, but kernels with hundreds of arguments make a real YAML parse slow: the per-kernel scalars are flat
        # `    .key: value` lines between two `  - .agpr_count` list heads
        blocks = re.split(r"\n  - (?=\.)", txt)
        names, vals = [], []
        for blk in blocks[1:]:
            m = re.search(r"\n\s+\.name:\s+(\S+)", "\n" + blk)
            if not m:
                continue
            kv = {}
            for key in _KEYS:
                mm = re.search(r"(?:^|\n)\s*\." + key + r":\s+(\d+)", blk)
                if mm:
                    kv[key] = int(mm.group(1))
            names.append(m.group(1))
            vals.append(kv)
        for n, kv in zip(_demangle(names), vals):
            res[re.sub(r"^void ", "", n)] = kv
    return res


if __name__ == "__main__":
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    table = kernel_resources()
    print(f"{'kernel':100s} vgpr agpr sgpr vspill sspill scratch  lds")
    for name in sorted(table):
        if pat in name:
            k = table[name]
            print(f"{name[:100]:100s} {k.get('vgpr_count', -1):4d} {k.get('agpr_count', -1):4d} {k.get('sgpr_count', -1):4d} "
                  f"{k.get('vgpr_spill_count', -1):6d} {k.get('sgpr_spill_count', -1):6d} {k.get('private_segment_fixed_size', -1):7d} "
                  f"{k.get('group_segment_fixed_size', -1):5d}")
