#!/usr/bin/env python3
"""Build-time guard on the kernels' register budgets and on the hand-counted asm loops.

  python tools/check_resources.py [file.hip ...]        (default: every csrc/*.hip; exit 1 on a violation)

1. Scratch.  Compiles each translation unit for gfx950 with -Rpass-analysis=kernel-resource-usage and fails when a kernel
   uses scratch memory (register spills) unless it is on the ALLOW list below -- the list names the cold instantiations that
   are known to spill, with the reason; every kernel a BASELINE configuration or a default plan selects must be clean.
3. Tickets (scan_tb.hip / scan_i8.hip): the asm global_atomic_add's destination register is untouched until the vmcnt(0) behind it.
2. Counted waits.  scan_w2_kernel (csrc/scan_w1.hip), gemm_big_kernel and gemm8_kernel pace their LDS reads / LDS-DMA with
   hand-counted s_waitcnt lgkmcnt(N) / vmcnt(N).  Those counts are only right while the compiler puts no SMEM load, scratch
   access or buffer instruction of its own between the counted statements: the device assembly of these kernels must have
   every s_load ahead of the first v_mfma, no scratch_* / buffer_* instruction at all, and a zero private segment.
"""
import concurrent.futures as cf
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "compressed-rag-suite_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only"]

# (regex on the demangled kernel name, why it may spill)
ALLOW = [
    (r"scan_f16_kernel<(512|640|768|896|1024), \d+, (16|32)", "threshold / compaction scan: only k > 48 (384: > 56, 768: > 40) on long streams reaches it"),
    (r"scan_f16_ring_kernel<(768|896|1024),", "ring variant of the same (CRS_SCAN_VARIANT, A/B runs only)"),
    (r"scan_i8_kernel<1024,", "int8 rows of 769-1024 elements: no BASELINE configuration; the digit planes of 64 queries x 1024 do not fit 256 registers"),
    (r"scan_i8_kernel<768, 32, 32, -1>", "int8 threshold kernel: k > 48 on long streams only"),
]
COUNTED = {"scan_w1.hip": ["scan_w2_kernel"], "enc_gemm_big.hip": ["gemm_big_kernel"], "enc_gemm8.hip": ["gemm8_kernel"]}
# 3. Tickets.  scan_tb_kernel / scan_i8_kernel draw their dynamic tiles with an asm global_atomic_add whose returned value is first
#    read behind a later s_waitcnt vmcnt(0) (the compiler does not count the asm's memory operation): between the atomic and that
#    wait no instruction may touch the destination register (a copy or spill there would carry garbage).
TICKETS = {"scan_tb.hip": "scan_tb_kernel", "scan_i8.hip": "scan_i8_kernel"}


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return out if len(out) == len(names) else names


def analyse(path):
    r = subprocess.run([HIPCC, *FLAGS, "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", os.devnull], capture_output=True, text=True)
    if r.returncode:
        return path, None, r.stderr[-2000:]
    txt = r.stderr
    names = re.findall(r"Function Name: (\S+)", txt)
    vg = [int(x) for x in re.findall(r" VGPRs: (\d+)", txt)]
    sc = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", txt)]
    rows = list(zip(demangle(names), vg, sc))
    asm_bad = []
    base = os.path.basename(path)
    if base in COUNTED:
        s = subprocess.run([HIPCC, *FLAGS, "-S", path, "-o", "-"], capture_output=True, text=True).stdout
        for kern in COUNTED[base]:
            for m in re.finditer(r"^(_Z\S*%s\S*):\n(.*?)s_endpgm" % kern, s, flags=re.S | re.M):
                body = m.group(2)
                first_mfma = body.find("v_mfma")
                late_sload = [ln for ln in body[first_mfma:].splitlines() if re.match(r"\s*s_load_", ln)] if first_mfma >= 0 else []
                bad = [ln.strip() for ln in body.splitlines() if re.match(r"\s*(scratch_|buffer_)", ln)]
                if late_sload or bad:
                    asm_bad.append((m.group(1), late_sload[:3] + bad[:3]))
            for m in re.finditer(r"\.name:\s+(\S*(?:%s)\S*)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)" % "|".join(COUNTED[base]), s):
                if int(m.group(2)):
                    asm_bad.append((m.group(1), ["private_segment_fixed_size %s" % m.group(2)]))
    if base in TICKETS:
        s = subprocess.run([HIPCC, *FLAGS, "-S", path, "-o", "-"], capture_output=True, text=True).stdout
        for m in re.finditer(r"^(_Z\S*%s\S*):\n(.*?)s_endpgm" % TICKETS[base], s, flags=re.S | re.M):
            lines = m.group(2).splitlines()
            for i, ln in enumerate(lines):
                am = re.match(r"\s*global_atomic_add\s+(v\d+),", ln)
                if not am:
                    continue
                reg = am.group(1)
                for nxt in lines[i + 1:]:
                    if re.match(r"\s*s_waitcnt\s+vmcnt\(0\)", nxt):
                        break
                    if re.search(r"\b%s\b" % reg, nxt) and not nxt.lstrip().startswith(";"):
                        asm_bad.append((m.group(1), ["ticket register %s touched before its wait: %s" % (reg, nxt.strip())]))
                        break
    return path, rows, asm_bad


def main(files):
    files = files or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    failures, allowed, n_kernels = [], [], 0
    with cf.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        for path, rows, extra in ex.map(analyse, files):
            if rows is None:
                failures.append(f"{os.path.basename(path)}: does not compile:\n{extra}")
                continue
            n_kernels += len(rows)
            for name, vgpr, scratch in rows:
                if scratch:
                    why = next((w for pat, w in ALLOW if re.search(pat, name)), None)
                    (allowed if why else failures).append(f"{os.path.basename(path)}: {name[:110]}: {scratch} B/lane of scratch, {vgpr} VGPRs"
                                                          + (f"   [allowed: {why}]" if why else ""))
            for kern, lines in extra:
                failures.append(f"{os.path.basename(path)}: {kern[:80]}: instruction the counted waits do not expect: {lines}")
    print(f"{n_kernels} kernels in {len(files)} files; {len(allowed)} on the allow list; {len(failures)} violation(s)")
    for a in allowed:
        print("  allowed  ", a)
    for f in failures:
        print("  VIOLATION", f)
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main([os.path.join(CSRC, a) if not os.path.exists(a) else a for a in sys.argv[1:]]))
