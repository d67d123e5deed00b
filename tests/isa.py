"""Reads the gfx950 code objects out of the shipped libsurs_hip.so: per-kernel metadata (registers, scratch, LDS) and disassembly.

Test infrastructure for tests/test_isa_pins.py (CPU only: llvm-objdump / llvm-readelf / clang-offload-bundler of /opt/rocm/lib/llvm).
The compiler hazards of NOTES R5.1 / R5.7 were found on hardware; what the compiler must NOT form is checked here in the shipped ISA."""
import os
import re
import shutil
import subprocess
import tempfile

import yaml

LLVM = os.environ.get("SURS_LLVM_BIN", "/opt/rocm/lib/llvm/bin")
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
SO = os.path.join(ROOT, "super-resolution-3d-human-shape-from-a-single-low-resolution-image_amd", "libsurs_hip.so")


def available():
    return all(os.path.exists(os.path.join(LLVM, t)) for t in ("llvm-objdump", "llvm-readelf")) and os.path.exists(SO)


def _run(args, **kw):
    return subprocess.run(args, check=True, capture_output=True, text=True, **kw).stdout


def code_objects(so=SO, workdir=None):
    """Extracts every gfx950 code object of `so` into workdir (llvm-objdump --offloading writes next to its input: the library is
    copied there first so that the package directory stays clean).  Returns the list of extracted files."""
    workdir = workdir or tempfile.mkdtemp(prefix="surs_isa_")
    local = os.path.join(workdir, os.path.basename(so))
    shutil.copyfile(so, local)
    _run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=workdir)
    return sorted(os.path.join(workdir, f) for f in os.listdir(workdir) if f.endswith("gfx950") and "hipv4" in f)


def kernel_metadata(co):
    """{mangled kernel name: metadata dict} from the code object's amdhsa.kernels note."""
    txt = _run([os.path.join(LLVM, "llvm-readelf"), "--notes", co])
    i = txt.find("amdhsa.kernels:")
    if i < 0:
        return {}
    j = txt.find("amdhsa.target:", i)
    doc = yaml.safe_load(txt[i:j if j > 0 else None])
    return {k[".name"]: k for k in doc["amdhsa.kernels"]}


_sym = re.compile(r"^(?:[0-9a-f]+ )?<([^>]+)>:\s*$")


def disassembly(co):
    """{symbol: [instruction text, ...]} of the code object's .text."""
    txt = _run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "--no-leading-addr", co])
    out, cur = {}, None
    for line in txt.splitlines():
        m = _sym.match(line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        s = line.strip()
        if cur is not None and s and not s.startswith("//"):
            cur.append(s.split("//")[0].strip())
    return out


def demangle(names):
    filt = shutil.which("c++filt") or os.path.join(LLVM, "llvm-cxxfilt")
    res = subprocess.run([filt], input="\n".join(names), capture_output=True, text=True, check=True).stdout.splitlines()
    return dict(zip(names, res))
