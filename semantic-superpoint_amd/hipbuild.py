"""Builds the HIP library in-tree (semantic-superpoint_amd/csrc/libssp_hip.so) for gfx950 and verifies the BINARY.
hipcc cross-compiles without a GPU; the .so is git-ignored but travels with gpurun snapshots.

Binary verification (`verify_binary`): the kernels that keep accumulators in FIXED accumulation registers behind inline
asm (conv_wino4.hip.h: a[0:255]) are only correct while the compiler never allocates an accumulation register or spills
in them - it does not know the registers are occupied.  The check disassembles the gfx950 code object embedded in the .so
that is actually shipped / loaded and requires, per such kernel, EXACTLY the accumulation-register instructions its
inline asm contains (any compiler-generated v_accvgpr_* / a-register operand changes the counts), no scratch and no
vector-register spills.  `build()` runs it after every compile and leaves a `<lib>.isa_ok` stamp (sha256 of the .so);
`lib.load_library()` re-verifies a library whose stamp is missing or stale, so the binary on the GPU box is covered too.
"""
import hashlib
import os
import re
import subprocess
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libssp_hip.so")
LLVM_BIN = "/opt/rocm/lib/llvm/bin"

# kernels with inline-asm-private accumulation registers -> the exact count of every instruction that may touch a0..a255:
# conv_wino4_kernel: per stage instance the first half (pairs 0..4 = 10 accumulators x 4 k-pairs) exists once per transform task kind
# (2 x 40) + pairs 5..7 (24) = 104; x 2 stage instances (the stage loop is unrolled by two) = 208 MFMAs on a[..]; two
# clear sites x 256 writes; the epilogue reads each of the 256 registers once in either wave role (2 x 256)
FIXED_AGPR_KERNELS = {
    "conv_wino4_kernel": {"v_mfma_f32_32x32x2_f32": 208, "v_accvgpr_write_b32": 512, "v_accvgpr_read_b32": 512},
    # wgrad_wino4_kernel: 16 accumulators x 4 K steps x 4 wave roles = 256 MFMAs on a[..] (the 2 x 4 x 4 = 32 of the two
    # vector-register accumulators carry no a-operand); one clear (256 writes); the epilogue reads every register once
    "wgrad_wino4_kernel": {"v_mfma_f32_32x32x2_f32": 256, "v_accvgpr_write_b32": 256, "v_accvgpr_read_b32": 256},
}
# kernels of the bf16 path that must compile without scratch memory (mangled-name fragments)
NO_SCRATCH_KERNELS = ("conv_bf16_ws_kernel", "wgrad_bf16_kernel")


class MissingTool(RuntimeError):
    """an LLVM binary utility needed by verify_binary is not installed"""


def _sources():
    out = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip") or f.endswith(".hip.h")]
    out.append(os.path.join(HERE, "..", "include", "ssp_hip.h"))
    return out


def source_id():
    """sha256 over the names and contents of every source file of the library (csrc/*.hip, csrc/*.hip.h, include/ssp_hip.h):
    the BUILD ID.  build() compiles it into the library (-DSSP_BUILD_ID, returned by ssp_build_id()), so a loaded binary can
    be tied to the checked-out sources by content instead of by modification time."""
    h = hashlib.sha256()
    for p in _sources():
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
        h.update(b"\0")
    extra = os.environ.get("SSP_HIPCC_EXTRA", "").strip()
    if extra:  # a library built with extra compiler flags (ablation macros) is NOT the build of the checked-out sources
        h.update(b"SSP_HIPCC_EXTRA\0" + extra.encode() + b"\0")
    return h.hexdigest()


def library_id(lib_path=LIB):
    """The build id compiled into `lib_path` (read from the file: the marker string SSP_BUILD_ID=<hex>), or None."""
    try:
        with open(lib_path, "rb") as f:
            data = f.read()
    except OSError:
        return None
    m = re.search(rb"SSP_BUILD_ID=([0-9a-f]{64})", data)
    return m.group(1).decode() if m else None


def _stale():
    """True when there is no library or its compiled-in build id differs from the hash of the checked-out sources."""
    if not os.path.exists(LIB):
        return True
    return library_id(LIB) != source_id()


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def _sha256(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def _tool(name):
    p = os.path.join(LLVM_BIN, name)
    if not os.path.exists(p):
        raise MissingTool("%s not found: cannot verify the accumulation-register contract of %s" % (p, LIB))
    return p


def disassemble(lib_path, workdir):
    """gfx950 code object of the fat binary inside `lib_path` -> (disassembly text, llvm-readelf --notes text)."""
    fat, co = os.path.join(workdir, "fat.bin"), os.path.join(workdir, "dev.co")
    subprocess.run([_tool("llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, lib_path, os.path.join(workdir, "unused.so")],
                   check=True, capture_output=True)
    subprocess.run([_tool("clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat,
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True, capture_output=True)
    dis = subprocess.run([_tool("llvm-objdump"), "-d", co], check=True, capture_output=True, text=True).stdout
    notes = subprocess.run([_tool("llvm-readelf"), "--notes", co], check=True, capture_output=True, text=True).stdout
    return dis, notes


_STORE_RE = re.compile(r"(?:buffer|global|flat|scratch)_store_dwordx[34]\s+(?:off,\s+|v\d+,\s+|v\[\d+:\d+\],\s+)?([va])\[(\d+):(\d+)\]")
_VALU_DST_RE = re.compile(r"(v_\S+)\s+(?:([va])\[(\d+):(\d+)\]|([va])(\d+))(?=[,\s]|$)")
_INSN_RE = re.compile(r"^[a-z][a-z0-9_]*(\s|$)")


def store_data_hazards(dis):
    """[(kernel, store, clobbering instruction)]: every 12 / 16-byte store of the disassembly `dis` whose NEXT INSTRUCTION is a
    vector-ALU instruction writing one of the store's data registers (vector or accumulation registers; the `off,` scratch form
    included; labels and other non-instruction lines between the two are skipped: a branch target does not hide a fall-through).
    The hardware reads the data of such a store for a few cycles after issue (one wait state).  LLVM's hazard recognizer pads this for
    the VALU instructions the COMPILER emits; what it cannot see is a VALU write inside INLINE ASM, which this code base uses
    everywhere (pk_math.hip.h): round 5 found conv_bf16_ws_kernel storing v7 << 16 in the last four lanes of every 16-lane row
    behind `buffer_store_dwordx4 v[6:9]; v_lshlrev_b32 v6, 16, v7` (the shift came from an asm helper).  New asm helpers that
    write a register a wide store has just read must keep the value alive across an s_nop (ssp_store_b128)."""
    out = []
    parts = re.split(r"^[0-9a-f]{16} <([^>]+)>:\n", dis, flags=re.M)
    for i in range(1, len(parts), 2):
        name = parts[i]
        # instruction lines only (llvm-objdump prints "<label>:" lines for branch targets inside a kernel)
        lines = [l.strip().split("//")[0].strip() for l in parts[i + 1].splitlines()]
        lines = [l for l in lines if l and _INSN_RE.match(l)]
        for k in range(len(lines) - 1):
            m = _STORE_RE.match(lines[k])
            if not m:
                continue
            bank, a, b = m.group(1), int(m.group(2)), int(m.group(3))
            mm = _VALU_DST_RE.match(lines[k + 1])
            if not mm or mm.group(1).startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
                continue
            if mm.group(2):
                dbank, lo, hi = mm.group(2), int(mm.group(3)), int(mm.group(4))
            else:
                dbank, lo, hi = mm.group(5), int(mm.group(6)), int(mm.group(6))
            if mm.group(1).startswith("v_accvgpr_write"):
                dbank = "a"
            if dbank == bank and lo <= b and hi >= a:
                out.append((name, lines[k], lines[k + 1]))
    return out


def verify_binary(lib_path=LIB, expected=None, require_all=None):
    """Raises RuntimeError if a fixed-accumulation-register kernel of `lib_path` contains any accumulation-register
    instruction beyond its inline asm, touches scratch, or spills vector registers, or if ANY kernel of the library overwrites
    the data of a wide store in the slot behind it (store_data_hazards).  Returns {kernel symbol: counts}.
    require_all (default: only for the in-tree library): every kernel family of the contract must be present - an A/B
    build of an older revision (SSP_HIP_LIB) may lack the newer kernels."""
    expected = expected if expected is not None else FIXED_AGPR_KERNELS
    if require_all is None:
        require_all = os.path.abspath(lib_path) == os.path.abspath(LIB)
    with tempfile.TemporaryDirectory(prefix="ssp_isa_", dir="/tmp") as tmp:
        dis, notes = disassemble(lib_path, tmp)
    parts = re.split(r"^[0-9a-f]{16} <([^>]+)>:\n", dis, flags=re.M)
    found, report = {k: 0 for k in expected}, {}
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1]
        fam = next((k for k in expected if k in name), None)
        if fam is None:
            continue
        counts = {}
        for line in body.splitlines():
            m = re.match(r"\s+(\S+)\s+(.*?)\s*//", line)
            if not m:
                continue
            op, args = m.group(1), m.group(2)
            if op.startswith("scratch_"):
                counts["scratch"] = counts.get("scratch", 0) + 1
            if re.search(r"\ba(\d+|\[\d+:\d+\])", args):
                counts[op] = counts.get(op, 0) + 1
        if counts != expected[fam]:
            raise RuntimeError("%s: accumulation-register instructions %s differ from the inline-asm contract %s - the compiler "
                               "allocated an accumulation register or spilled in a kernel whose a[0:255] are private"
                               % (name, counts, expected[fam]))
        found[fam] += 1
        report[name] = counts
    for fam, n in found.items():
        if n == 0 and require_all:
            raise RuntimeError("no %s instance found in %s" % (fam, lib_path))
    hazards = store_data_hazards(dis)
    if hazards:
        raise RuntimeError("store-data hazard (a vector instruction overwrites a data register of a 12 / 16-byte store in the next "
                           "slot; gfx950 needs one wait state, hipcc does not insert it - use ssp_store_b128, csrc/pk_math.hip.h): "
                           + "; ".join("%s: %s -> %s" % h for h in hazards[:8]) + (" ... %d in all" % len(hazards) if len(hazards) > 8 else ""))
    # kernel descriptors: no private segment, no vector-register spills
    seen_notes = set()
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        m = re.search(r"\.symbol:\s+(\S+?)\.kd", blk)  # (the kernel's own symbol: `.name:` also occurs in the argument metadata)
        if not m or not any(k in m.group(1) for k in expected):
            continue
        seen_notes.add(next(k for k in expected if k in m.group(1)))
        seg = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        agpr = int(re.search(r"^\s*(\d+)", blk).group(1))
        if seg != 0 or spill != 0 or agpr != 256:
            raise RuntimeError("%s: private segment %d B, %d spilled vector registers, %d accumulation registers reserved "
                               "(need 0 / 0 / 256)" % (m.group(1), seg, spill, agpr))
    for fam, n in found.items():
        if n > 0 and fam not in seen_notes:
            raise RuntimeError("%s is in the disassembly of %s but its kernel descriptor was not found in the notes: the "
                               "spill / private-segment check did not run" % (fam, lib_path))
    # the bf16 path's hot kernels sit at the register limit of their launch bounds: a spilled register puts scratch loads
    # (and their s_waitcnt vmcnt) into the staging loop - 12 bytes of scratch measured +16 % cycles on the 3x3 forward
    for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        m = re.search(r"\.symbol:\s+(\S+?)\.kd", blk)
        if not m or not any(k in m.group(1) for k in NO_SCRATCH_KERNELS):
            continue
        seg = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1))
        spill = int(re.search(r"\.vgpr_spill_count:\s+(\d+)", blk).group(1))
        if seg != 0 or spill != 0:
            raise RuntimeError("%s: private segment %d B, %d spilled vector registers (this kernel must not touch scratch)"
                               % (m.group(1), seg, spill))
        report[m.group(1)] = {"scratch": 0}
    return report


def stamp_path(lib_path=LIB):
    return lib_path + ".isa_ok"


def verified(lib_path=LIB):
    """True iff `lib_path` carries a stamp written by verify_and_stamp for exactly this binary."""
    try:
        return open(stamp_path(lib_path)).read().strip() == _sha256(lib_path)
    except OSError:
        return False


def verify_and_stamp(lib_path=LIB):
    verify_binary(lib_path)
    with open(stamp_path(lib_path), "w") as f:
        f.write(_sha256(lib_path) + "\n")


def build(force=False, verbose=False):
    """Compile csrc/ssp.hip -> csrc/libssp_hip.so (gfx950), verify the binary.  Returns the library path."""
    if not force and not _stale():
        if not verified():
            verify_and_stamp()
        return LIB
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
           "-DSSP_BUILD_ID=\"%s\"" % source_id(), "ssp.hip", "-o", "libssp_hip.so"] + os.environ.get("SSP_HIPCC_EXTRA", "").split()
    r = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0 or verbose:
        print(r.stdout)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed (%d): %s" % (r.returncode, " ".join(cmd)))
    try:
        verify_and_stamp()
    except Exception:
        # a library that violates the register contract must never be loadable by accident
        os.replace(LIB, LIB + ".rejected")
        raise
    return LIB


def build_locked():
    """build() when stale, with an exclusive file lock so that the ranks of one node do not compile concurrently."""
    if not _stale() and verified():
        return LIB
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            return build()  # re-checks _stale() / the stamp under the lock
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


if __name__ == "__main__":
    print(build(force=True, verbose=True))
