"""Removes the development-only probe branches (#ifdef Y3D_W3_* / Y3D_PROBE_* / Y3D_STAGGER ... #else ... #endif, TRC(...) stamps) from a
kernel source: the product file carries none of them, the probe copy under tools/probe/ keeps them.
    python tools/probe/strip_probes.py <src> <dst>"""
import re
import sys

PROBE = re.compile(r"Y3D_W3_|Y3D_PROBE|Y3D_STAGGER")


def strip(text):
    out, stack = [], []
    for l in text.split("\n"):
        s = l.strip()
        if s.startswith(("#ifdef", "#ifndef", "#if ")):
            if PROBE.search(s):
                if s.startswith("#ifdef"):
                    val = False
                elif s.startswith("#ifndef"):
                    val = True
                else:
                    assert re.fullmatch(r"#if (!defined\(\w+\))( && !defined\(\w+\))*", s.split("//")[0].strip()), s
                    val = True
                stack.append(["probe", val])
                continue
            stack.append(["keep", True])
        elif s.startswith("#else") and stack and stack[-1][0] == "probe":
            stack[-1][1] = not stack[-1][1]
            continue
        elif s.startswith("#endif"):
            if stack.pop()[0] == "probe":
                continue
        if all(v for _, v in stack):
            out.append(l)
    src = "\n".join(out)
    src = re.sub(r"\n *TRC\([^\n]*\);", "", src)
    src = re.sub(r"\n#define (TRC|W3_BARRIER)\([^\n]*", "", src)
    src = src.replace("W3_BARRIER()", "__builtin_amdgcn_s_barrier()")
    return src


if __name__ == "__main__":
    open(sys.argv[2], "w").write(strip(open(sys.argv[1]).read()))
