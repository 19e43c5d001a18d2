"""Generates tests/golden/reference_constants.json: every number of the reference that the dynamics path takes as a literal, read from
the reference's own files and EVALUATED BY THE COMPILER (so `1./3.` or `10e3` is the double the reference's build gets).  Run in the build
container (needs /root/reference and g++):   python tests/golden/make_reference_constants.py

  physical   model/constants.hpp (includes nothing): the header is #included in a generated translation unit, every `const double NAME`
             it declares is printed with %a
  enums      model/enums.hpp (includes nothing): likewise, every enumerator of every `enum class` as an int
  pi         contrib/bamg/include/OppositeAngle.h:4, the PI macro FE.cpp's `PI` resolves to
  members    model/finiteelement.hpp:549-550 `double const days_in_sec / years_in_sec` (in-class initialisers: the initialiser text is
             compiled, the header itself needs Boost)
  options    model/options.cpp: every ("section.name", po::value<T>()->default_value( EXPR ) that is not commented out; numeric and bool
             EXPRs are compiled as `T v = EXPR;`, string defaults are kept as text

Only the JSON travels (the reference does not exist on the GPU box); tests/test_reference_constants.py compares oracle/dyn_ref.c's
constants, include/nxs_dyn.h's enums and nxs_dyn_default_params / ref_default_params / forcing.default_params with it.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("NXS_REFERENCE", "/root/reference")


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return "\n".join(line.split("//")[0] for line in text.split("\n"))


def main():
    constants = strip_comments(open(os.path.join(REF, "model", "constants.hpp")).read())
    enums = strip_comments(open(os.path.join(REF, "model", "enums.hpp")).read())
    options = strip_comments(open(os.path.join(REF, "model", "options.cpp")).read())
    fehpp = strip_comments(open(os.path.join(REF, "model", "finiteelement.hpp")).read())

    phys = re.findall(r"const\s+double\s+(\w+)\s*=", constants)
    enum_items = []
    for m in re.finditer(r"enum\s+class\s+(\w+)\s*\{(.*?)\}", enums, flags=re.S):
        space = re.findall(r"namespace\s+(\w+)", enums[:m.start()])[-1]   # (setup / schemes: the innermost namespace opened last)
        for item in m.group(2).split(","):
            name = item.split("=")[0].strip()
            if name:
                enum_items.append((space, m.group(1), name))
    members = dict(re.findall(r"double\s+const\s+(days_in_sec|years_in_sec)\s*=\s*([^;]+);", fehpp))
    opts = []
    for m in re.finditer(r'\(\s*"([\w.\-]+)"\s*,\s*po::value<\s*([\w:]+)\s*>\s*\(\s*\)\s*->\s*default_value\s*\(', options):
        depth, j = 1, m.end()
        while depth:                                   # the argument up to its matching parenthesis
            depth += {"(": 1, ")": -1}.get(options[j], 0)
            j += 1
        opts.append((m.group(1), m.group(2), options[m.end():j - 1].strip()))
    numeric = [(n, t, e) for n, t, e in opts if t in ("double", "int", "bool")]
    strings = {n: e.strip().strip('"') for n, t, e in opts if t == "std::string"}

    lines = ['#include <cstdio>', f'#include "{REF}/model/constants.hpp"', f'#include "{REF}/model/enums.hpp"',
             f'#include "{REF}/contrib/bamg/include/OppositeAngle.h"', "int main() {"]
    for n in phys:
        lines.append(f'  std::printf("physical {n} %a\\n", (double)physical::{n});')
    for sp, e, n in enum_items:
        lines.append(f'  std::printf("enum {e}.{n} %d\\n", (int)Nextsim::{sp}::{e}::{n});')
    lines.append('  std::printf("pi PI %a\\n", (double)(PI));')
    lines.append(f'  {{ double const days_in_sec = {members["days_in_sec"]}; double const years_in_sec = {members["years_in_sec"]};')
    lines.append('    std::printf("member days_in_sec %a\\nmember years_in_sec %a\\n", days_in_sec, years_in_sec); }')
    for n, t, e in numeric:
        fmt = "%a" if t == "double" else "%d"
        cast = "(double)" if t == "double" else "(int)"
        lines.append(f'  {{ {t} v = {e}; std::printf("option {n} {t} {fmt}\\n", {cast}v); }}')
    lines.append("  return 0; }")

    with tempfile.TemporaryDirectory() as tmp:
        src, exe = os.path.join(tmp, "c.cpp"), os.path.join(tmp, "c")
        open(src, "w").write("\n".join(lines) + "\n")
        subprocess.check_call(["g++", "-std=c++14", "-O0", "-ffp-contract=off", src, "-o", exe])
        rows = subprocess.check_output([exe], text=True).split("\n")

    out = {"_generated_by": "tests/golden/make_reference_constants.py (values printed by a g++-compiled translation unit that includes the reference's headers / "
                            "default_value expressions; doubles as C99 hex floats, `value` is the same number in decimal for the reader)",
           "physical": {}, "enums": {}, "pi": None, "members": {}, "options": {}, "string_options": strings}

    def num(h):
        return {"hex": h, "value": float.fromhex(h)}
    for row in rows:
        f = row.split()
        if not f:
            continue
        if f[0] == "physical":
            out["physical"][f[1]] = num(f[2])
        elif f[0] == "enum":
            e, n = f[1].split(".")
            out["enums"].setdefault(e, {})[n] = int(f[2])
        elif f[0] == "pi":
            out["pi"] = num(f[2])
        elif f[0] == "member":
            out["members"][f[1]] = num(f[2])
        elif f[0] == "option":
            out["options"][f[1]] = {"type": f[2], **(num(f[3]) if f[2] == "double" else {"value": int(f[3])})}
    dst = os.path.join(HERE, "reference_constants.json")
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(f"{dst}: {len(out['physical'])} physical constants, {sum(len(v) for v in out['enums'].values())} enumerators, "
          f"{len(out['options'])} numeric option defaults, {len(strings)} string option defaults")


if __name__ == "__main__":
    sys.exit(main())
