#!/usr/bin/env python3
"""Static checker of julia/CMXExt.jl — the reference-side binding of libcmx.so.

Julia is not in the build image, so the binding cannot be executed here.  This tool reads three texts instead and makes them agree:

  * the reference's own `struct` definitions (/root/reference/src/parameters/*.jl, src/AerosolModel.jl, src/Quadrature.jl) — skipped,
    with a note, when /root/reference is absent (the GPU box);
  * include/cmx.h: parameter structs, CMX_ASSERT_PARAM_STRUCT_SIZES, the `#define` flags, every `cmx_*` prototype;
  * julia/CMXExt.jl: `Cmx*` mirror structs (`# == cmx_…`), the DIRECT_LAYOUT table, every field access on a type-annotated argument,
    the flag constants, the version constants, every `ccall(_fn("cmx_…", FT), …)`.

It fails on: a field the reference struct does not have; a nested access `x.a.b` on an annotated argument (the file's rule: one level per
method, so that this check is sound); a mirror struct whose field names / order / array lengths / size differ from the C struct; a
DIRECT_LAYOUT pair whose transliterated reference field list differs from the C struct; a `CMP.X` / `BMT.X` / `AM.X` / `AA.X` / `QUAD.X` name
the reference does not define; a `ccall` whose argument-type tuple or argument count differs from the C prototype; a flag constant whose
value differs from the header; an entry-point family of the header without a binding; unbalanced delimiters or `end`s.

    python tools/check_julia_shim.py            # prints a summary, exit code 1 on any finding
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
HEADER = REPO / "include" / "cmx.h"
SHIM = REPO / "julia" / "CMXExt.jl"
REFERENCE = Path("/root/reference")

IDENT = r"[^\W\d][\w!]*"          # Julia identifier (unicode letters, subscripts are \w)

# ---------------------------------------------------------------------------------------------------------------------------------
# Julia source helpers
# ---------------------------------------------------------------------------------------------------------------------------------


def strip_julia(text: str, keep_strings: bool = False) -> str:
    """Blank out comments, docstrings and (unless keep_strings) string literals, preserving length and newlines."""
    out = []
    i, n = 0, len(text)
    while i < n:
        c = text[i]
        if text.startswith('"""', i):
            j = text.find('"""', i + 3)
            j = n if j < 0 else j + 3
            seg = text[i:j]
            out.append(re.sub(r"[^\n]", " ", seg))
            i = j
        elif c == '"':
            j = i + 1
            while j < n and text[j] != '"':
                j += 2 if text[j] == "\\" else 1
            seg = text[i:j + 1]
            out.append(seg if keep_strings else '"' + re.sub(r"[^\n]", " ", seg[1:-1]) + '"')
            i = j + 1
        elif text.startswith("#=", i):
            j = text.find("=#", i + 2)
            j = n if j < 0 else j + 2
            out.append(re.sub(r"[^\n]", " ", text[i:j]))
            i = j
        elif c == "#":
            j = text.find("\n", i)
            j = n if j < 0 else j
            out.append(" " * (j - i))
            i = j
        elif c == "'" and i + 2 < n and (text[i + 2] == "'" or (text[i + 1] == "\\" and i + 3 < n and text[i + 3] == "'")):
            j = i + (3 if text[i + 2] == "'" else 4)
            out.append(" " * (j - i))
            i = j
        else:
            out.append(c)
            i += 1
    return "".join(out)


def parse_julia_structs(text: str) -> dict:
    """{name: [(field, type-or-None), …]} for every `struct` of a Julia source (docstrings / comments removed first)."""
    src = strip_julia(text)
    lines = src.split("\n")
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(rf"^\s*(?:Base\.)?(?:@kwdef\s+)?(?:mutable\s+)?struct\s+({IDENT})", lines[i])
        if not m:
            i += 1
            continue
        name = m.group(1)
        hdr = lines[i][m.end():]
        depth = hdr.count("{") - hdr.count("}")
        while depth > 0:
            i += 1
            hdr += " " + lines[i]
            depth += lines[i].count("{") - lines[i].count("}")
        fields = []
        if re.search(r"\bend\s*$", hdr):
            out[name] = fields
            i += 1
            continue
        i += 1
        nest = 0
        while i < len(lines):
            t = lines[i].strip()
            if not t or t.startswith('"'):
                i += 1
                continue
            if re.match(r"^(function|for|if|let|begin|while)\b", t):
                nest += 1
            elif t == "end" or t.startswith("end "):
                if nest == 0:
                    break
                nest -= 1
            elif nest == 0:
                fm = re.match(rf"^({IDENT})\s*(?:::\s*([^=]+?))?\s*(?:=.*)?$", t)
                if fm and not re.match(rf"^{IDENT}\s*\(", t):
                    fields.append((fm.group(1), (fm.group(2) or "").strip() or None))
            i += 1
        out[name] = fields
        i += 1
    return out


GREEK = {"α": "alpha", "β": "beta", "γ": "gamma", "Δ": "delta", "δ": "delta", "ϵ": "eps", "ε": "eps", "κ": "kappa", "λ": "lambda", "μ": "mu",
         "ν": "nu", "ρ": "rho", "σ": "sigma", "τ": "tau", "ϕ": "phi", "φ": "phi", "χ": "chi", "π": "pi"}
SUBSCRIPT = {"₀": "0", "₁": "1", "₂": "2", "₃": "3", "₄": "4", "₅": "5", "₆": "6", "₇": "7", "₈": "8", "₉": "9", "ᵢ": "_i", "ᵥ": "_v", "ₐ": "_a"}


def translit(name: str) -> str:
    """ASCII transliteration of a Julia field name the way include/cmx.h spells it: νc → nu_c, ρ0 → rho_0, Δa_w_min → delta_a_w_min,
    c₁ → c1, Sᵢ_max → S_i_max, ρᵢ → rho_i, b_ρ → b_rho."""
    out = []
    for k, ch in enumerate(name):
        if ch in GREEK:
            out.append(GREEK[ch])
            nxt = name[k + 1] if k + 1 < len(name) else ""
            if nxt and nxt != "_" and nxt not in SUBSCRIPT:
                out.append("_")
        elif ch in SUBSCRIPT:
            out.append(SUBSCRIPT[ch])
        else:
            out.append(ch)
    return "".join(out)


# ---------------------------------------------------------------------------------------------------------------------------------
# include/cmx.h
# ---------------------------------------------------------------------------------------------------------------------------------


def strip_c_comments(text: str) -> str:
    return re.sub(r"/\*.*?\*/", lambda m: re.sub(r"[^\n]", " ", m.group(0)), text, flags=re.S)


class Header:
    def __init__(self, path: Path = HEADER):
        raw = path.read_text()
        self.text = strip_c_comments(raw)
        self.defines = {}
        for m in re.finditer(r"^#define\s+(CMX_\w+)\s+(.+?)\s*$", self.text, flags=re.M):
            v = self._int(m.group(2))
            if v is not None:
                self.defines[m.group(1)] = v
        for m in re.finditer(r"typedef enum \w+ \{(.*?)\}", self.text, flags=re.S):
            k = 0
            for item in m.group(1).split(","):
                item = item.strip()
                if not item:
                    continue
                if "=" in item:
                    nm, val = [x.strip() for x in item.split("=")]
                    k = int(val, 0)
                else:
                    nm = item
                self.defines[nm] = k
                k += 1
        self.structs = self._structs()
        self.sizes = self._sizes()
        self.protos = self._protos()

    @staticmethod
    def _int(expr: str):
        e = expr.strip()
        if e.startswith("(") and e.endswith(")"):
            e = e[1:-1].strip()
        m = re.fullmatch(r"(\d+)u?\s*<<\s*(\d+)", e)
        if m:
            return int(m.group(1)) << int(m.group(2))
        m = re.fullmatch(r"(\d+)u?", e)
        return int(m.group(1)) if m else None

    def _structs(self):
        """{base name (no _f32/_f64): [(field, kind, arg)]}: kind 'ft' (arg = array length or 0), 'i32', 'struct' (arg = base name)."""
        body = self.text[self.text.index("#define CMX_DECLARE_PARAM_STRUCTS"):self.text.index("#define CMX_ARG_MAX_MODES")]
        body = body.replace("\\\n", "\n")
        out = {}
        for m in re.finditer(r"typedef struct (cmx_\w+?)_##SFX \{(.*?)\}\s*\1_##SFX;", body, flags=re.S):
            fields = []
            for decl in m.group(2).split(";"):
                decl = " ".join(decl.split())
                if not decl:
                    continue
                dm = re.match(r"^(FT|int32_t|cmx_\w+?_##SFX)\s+(.*)$", decl)
                if not dm:
                    raise ValueError(f"cannot parse member '{decl}' of {m.group(1)}")
                ty = dm.group(1)
                for var in dm.group(2).split(","):
                    var = var.strip()
                    am = re.match(r"^(\w+)\[(\w+)\]$", var)
                    nm, alen = (am.group(1), am.group(2)) if am else (var, None)
                    if alen is not None:
                        alen = int(alen) if alen.isdigit() else self.defines[alen]
                    if ty == "FT":
                        fields.append((nm, "ft", alen or 0))
                    elif ty == "int32_t":
                        fields.append((nm, "i32", 0))
                    else:
                        fields.append((nm, "struct", (ty[:-len("_##SFX")], alen or 0)))
            out[m.group(1)] = fields
        return out

    def _sizes(self):
        out = {}
        for m in re.finditer(r"sizeof\((cmx_\w+?)_##SFX\) == (?:(\d+) \+ )?(\d+) \* sizeof\(FT\)", self.text):
            out[m.group(1)] = (int(m.group(2) or 0), int(m.group(3)))
        return out

    def size_of(self, name: str):
        """(integer-header bytes, number of FT) of a struct, by flattening its members."""
        b = n = 0
        for _, kind, arg in self.structs[name]:
            if kind == "ft":
                n += max(arg, 1)
            elif kind == "i32":
                b += 4
            else:
                sb, sn = self.size_of(arg[0])
                b += sb * max(arg[1], 1)
                n += sn * max(arg[1], 1)
        return b, n

    def leaves(self, name: str):
        out = []
        for nm, kind, arg in self.structs[name]:
            if kind == "struct":
                for _ in range(max(arg[1], 1)):
                    out += self.leaves(arg[0])
            else:
                out.append((nm, kind, arg))
        return out

    def _protos(self):
        """{family name (no _f32/_f64): [julia type string per parameter]} from the `_f32` prototypes."""
        out = {}
        for m in re.finditer(r"int32_t\s+(cmx_\w+)\s*\(([^;{]*?)\)\s*;", self.text, flags=re.S):
            name, params = m.group(1), " ".join(m.group(2).split())
            if not name.endswith("_f32"):
                if name.endswith("_f64") or name in ("cmx_version",):
                    continue
                if name.startswith("cmx_lean_eval"):
                    continue
            types = []
            for p in params.split(","):
                p = p.strip()
                if p == "void":
                    continue
                types.append(self._jl_type(p))
            out[name[:-4] if name.endswith("_f32") else name] = types
        return out

    @staticmethod
    def _jl_type(p: str) -> str:
        p = re.sub(r"\[\w*\]$", "*", p.strip())          # `T *const out[N]` is `T *const *out`
        stars = p.count("*")
        base = re.sub(r"\bconst\b", "", p.split("*")[0] if stars else " ".join(p.split()[:-1]))
        base = " ".join(base.split())
        if stars and not base:
            raise ValueError(p)
        if stars == 0:
            return {"uint32_t": "UInt32", "int64_t": "Int64", "int32_t": "Int32", "float": "FT", "double": "Float64"}[base]
        # (the `_f32` prototypes are the ones parsed: `float` is FT, a `double` there is a genuine Float64)
        prim = {"float": "FT", "double": "Float64", "int64_t": "Int64", "void": "Cvoid"}.get(base.split()[0] if base else "")
        if base.startswith("cmx_"):
            prim = "Cvoid"
        if prim is None:
            raise ValueError(f"unknown C parameter type '{p}'")
        t = prim
        for _ in range(stars):
            t = f"Ptr{{{t}}}"
        return t


# ---------------------------------------------------------------------------------------------------------------------------------
# julia/CMXExt.jl
# ---------------------------------------------------------------------------------------------------------------------------------


def split_top(s: str, sep: str = ","):
    """Split at top-level separators (outside (), [], {})."""
    parts, depth, cur = [], 0, []
    for ch in s:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == sep and depth == 0:
            parts.append("".join(cur).strip())
            cur = []
        else:
            cur.append(ch)
    tail = "".join(cur).strip()
    if tail:
        parts.append(tail)
    return parts


def matching(s: str, i: int) -> int:
    """Index of the delimiter closing the one at s[i]."""
    pairs = {"(": ")", "[": "]", "{": "}"}
    depth = 0
    for j in range(i, len(s)):
        if s[j] in pairs:
            depth += 1
        elif s[j] in pairs.values():
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced delimiter")


class Shim:
    def __init__(self, path: Path = SHIM, text: str = None):
        self.raw = path.read_text(encoding="utf-8") if text is None else text
        self.src = strip_julia(self.raw, keep_strings=True)
        self.nostr = strip_julia(self.raw)
        self.mirrors = self._mirrors()
        self.direct = self._direct()
        self.consts = self._consts()
        self.functions = self._functions()
        self.ccalls = self._ccalls()

    def _mirrors(self):
        """{julia struct: (c struct base name, [(field, type)])} — the `# == cmx_…` tag sits on the `struct` line."""
        out = {}
        structs = parse_julia_structs(self.raw)
        for m in re.finditer(rf"^struct\s+({IDENT})[^\n#]*#\s*==\s*(cmx_\w+)", self.raw, flags=re.M):
            out[m.group(1)] = (m.group(2), structs[m.group(1)])
        untagged = [s for s in structs if s.startswith("Cmx") and s not in out]
        if untagged:
            raise ValueError(f"mirror structs without a '# == cmx_…' tag: {untagged}")
        return out

    def _direct(self):
        i = self.nostr.index("const DIRECT_LAYOUT = (")
        j = matching(self.nostr, self.nostr.index("(", i))
        rows = []
        for row in split_top(self.nostr[self.nostr.index("(", i) + 1:j]):
            row = row.strip()
            if not row:
                continue
            parts = split_top(row[1:matching(row, 0)])
            ref = parts[0].strip()
            nested = {}
            inner = parts[1].strip()
            for pr in split_top(inner[1:-1]):
                pm = re.match(rf"^:({IDENT})\s*=>\s*([\w.]+)$", pr.strip())
                if pm:
                    nested[pm.group(1)] = pm.group(2)
            rows.append((ref, nested, parts[2].strip().lstrip(":")))
        return rows

    def _consts(self):
        out = {}
        for m in re.finditer(r"^const\s+(CMX_\w+)\s*=\s*(.+?)\s*$", self.nostr, flags=re.M):
            e = m.group(2)
            mm = re.fullmatch(r"UInt32\((\d+)\)\s*<<\s*(\d+)", e)
            if mm:
                out[m.group(1)] = int(mm.group(1)) << int(mm.group(2))
                continue
            mm = re.fullmatch(r"(?:UInt32\()?(\d+)\)?", e)
            if mm:
                out[m.group(1)] = int(mm.group(1))
        return out

    def _functions(self):
        """[(name, {arg: type}, body text, line)] for long-form (`function … end`) and short-form (`f(args) = expr`) definitions."""
        src = self.nostr
        out = []
        # long form
        for m in re.finditer(rf"^function\s+({IDENT}|\(\s*{IDENT}\s*::[^)]*\))\s*\(", src, flags=re.M):
            p0 = m.end() - 1
            p1 = matching(src, p0)
            args = src[p0 + 1:p1]
            end = re.search(r"^end\b", src[p1:], flags=re.M)
            body = src[p1:p1 + end.start()] if end else src[p1:]
            out.append((m.group(1), self._arg_types(args), body, src[:m.start()].count("\n") + 1))
        # short form: name(args) [where {...}] = body (up to the next line that does not continue the expression)
        for m in re.finditer(rf"^({IDENT})\(", src, flags=re.M):
            if m.group(1) in ("function", "struct", "const", "import", "module", "end", "for", "if", "return", "error"):
                continue
            p0 = m.end() - 1
            try:
                p1 = matching(src, p0)
            except ValueError:
                continue
            rest = src[p1 + 1:]
            mm = re.match(r"^(\s*where\s*\{[^}]*\})?\s*=(?!=)", rest)
            if not mm:
                continue
            start = p1 + 1 + mm.end()
            # body: to the end of the statement = until a newline at delimiter depth 0 that is not followed by an indented continuation
            k, depth = start, 0
            while k < len(src):
                ch = src[k]
                if ch in "([{":
                    depth += 1
                elif ch in ")]}":
                    depth -= 1
                elif ch == "\n" and depth == 0:
                    nxt = src[k + 1:k + 2]
                    if nxt not in (" ", "\t"):
                        break
                k += 1
            out.append((m.group(1), self._arg_types(src[p0 + 1:p1]), src[start:k], src[:m.start()].count("\n") + 1))
        return out

    @staticmethod
    def _arg_types(args: str):
        types = {}
        for a in split_top(args.replace(";", ",")):
            am = re.match(rf"^({IDENT})\s*::\s*([\w.]+)", a.strip())
            if am:
                types[am.group(1)] = am.group(2)
        return types

    def _ccalls(self):
        out = []
        for m in re.finditer(r'ccall\(_fn\("(cmx_\w+)",\s*FT\)', self.src):
            p0 = self.src.index("(", m.start())
            p1 = matching(self.src, p0)
            parts = split_top(self.src[p0 + 1:p1])
            ret, types = parts[1], split_top(parts[2].strip()[1:-1])
            out.append((m.group(1), ret, [t.strip() for t in types if t.strip()], parts[3:], self.src[:m.start()].count("\n") + 1))
        return out


# ---------------------------------------------------------------------------------------------------------------------------------
# reference
# ---------------------------------------------------------------------------------------------------------------------------------


class Reference:
    MODULE_FILES = {
        "CMP": ["src/parameters/*.jl"],
        "BMT": ["src/BulkMicrophysicsTendencies.jl"],
        "AM": ["src/AerosolModel.jl"],
        "AA": ["src/AerosolActivation.jl"],
        "QUAD": ["src/Quadrature.jl"],
    }
    # accessors of Thermodynamics.jl's parameter set (un-vendored).  The first group is used by the reference's own sources; the second
    # group is Thermodynamics.jl's public accessor API for the same struct (the constants its saturation-vapour-pressure and latent-heat
    # formulas read), which the reference reaches only through TD functions.
    TDP_USED_BY_REFERENCE = {"T_freeze", "R_v", "R_d", "cp_d", "cp_l", "cv_l", "LH_v0", "LH_s0", "q_min", "grav", "Rv_over_Rd", "ThermodynamicsParameters"}
    TDP_THERMODYNAMICS_OWN = {"cp_v", "cp_i", "T_0", "T_triple", "press_triple"}

    def __init__(self, root: Path = REFERENCE):
        self.root = root
        self.structs = {}
        self.text = {}
        for mod, globs in self.MODULE_FILES.items():
            txt = ""
            for g in globs:
                for f in sorted(root.glob(g)):
                    t = f.read_text(encoding="utf-8")
                    txt += "\n" + t
                    for k, v in parse_julia_structs(t).items():
                        self.structs[k] = v
            self.text[mod] = strip_julia(txt)
        opt = strip_julia((root / "src/parameters/Microphysics1MOptions.jl").read_text(encoding="utf-8"))
        i = opt.index("microphysics_1m_process_params(td::CP.ParamDict")
        blk = opt[i:opt.index("\n)\n", i)]
        self.process_param_keys = set(re.findall(rf"^\s*({IDENT})\s*=\s*process_params_for", blk, flags=re.M))
        self.process_param_inner = set(re.findall(rf"=>\s*:({IDENT})", opt)) | set(re.findall(rf"\(;\s*({IDENT})\s*=", opt)) | \
            set(re.findall(rf",\s*({IDENT})\s*=\s*Frostenberg2023", opt))

    def defines(self, mod: str, name: str) -> bool:
        t = self.text[mod]
        return bool(re.search(rf"\b(struct|abstract type|function)\s+{re.escape(name)}\b", t)
                    or re.search(rf"^\s*(?:@inline\s+)?{re.escape(name)}\(", t, flags=re.M)
                    or re.search(rf"^\s*const\s+{re.escape(name)}\b", t, flags=re.M))


# ---------------------------------------------------------------------------------------------------------------------------------
# the checks
# ---------------------------------------------------------------------------------------------------------------------------------


def lint(shim: Shim, findings):
    s = shim.nostr
    # delimiters
    stack = []
    pairs = {")": "(", "]": "[", "}": "{"}
    line = 1
    for ch in s:
        if ch == "\n":
            line += 1
        elif ch in "([{":
            stack.append((ch, line))
        elif ch in pairs:
            if not stack or stack[-1][0] != pairs[ch]:
                findings.append(f"lint: unbalanced '{ch}' at line {line}")
                return
            stack.pop()
    if stack:
        findings.append(f"lint: unclosed '{stack[-1][0]}' opened at line {stack[-1][1]}")
    # block openers vs `end`, counted outside (), [], {} only (a comprehension's `for` / `if` needs no `end`; string contents and
    # comments are already blank)
    depth = dl = 0
    for m in re.finditer(r"[()\[\]{}]|(?<![\w.:!@])(?:module|struct|function|for|while|if|let|begin|do|try|quote|end)(?![\w!])", s):
        t = m.group(0)
        if t in "([{":
            dl += 1
        elif t in ")]}":
            dl -= 1
        elif dl == 0:
            depth += -1 if t == "end" else 1
            if depth < 0:
                findings.append(f"lint: an `end` without an opener at line {s[:m.start()].count(chr(10)) + 1}")
                return
    if depth != 0:
        findings.append(f"lint: {depth} block(s) not closed by `end`")
    if "\t" in shim.raw:
        findings.append("lint: tab character")


def check_mirrors(shim: Shim, hdr: Header, findings):
    for jname, (cname, jfields) in shim.mirrors.items():
        if cname not in hdr.structs:
            findings.append(f"{jname}: C struct {cname} does not exist in include/cmx.h")
            continue
        cfields = hdr.structs[cname]
        if [f for f, _ in jfields] != [f for f, _, _ in cfields]:
            findings.append(f"{jname} == {cname}: field names/order differ\n    julia: {[f for f, _ in jfields]}\n    C:     {[f for f, _, _ in cfields]}")
            continue
        for (jf, jt), (cf, kind, arg) in zip(jfields, cfields):
            jt = jt or ""
            if kind == "i32" and jt != "Int32":
                findings.append(f"{jname}.{jf}: C int32_t, Julia {jt}")
            if kind == "ft":
                if arg == 0 and jt not in ("FT",):
                    findings.append(f"{jname}.{jf}: C scalar FT, Julia {jt}")
                if arg > 0 and jt.replace(" ", "") != f"NTuple{{{arg},FT}}":
                    findings.append(f"{jname}.{jf}: C FT[{arg}], Julia {jt}")
            if kind == "struct":
                sub, alen = arg
                mm = re.match(r"^(?:NTuple\{(\d+),\s*)?(Cmx\w+)\{FT\}\}?$", jt)
                if mm:      # concrete mirror member: must be the mirror of that C struct
                    if shim.mirrors.get(mm.group(2), (None,))[0] != sub:
                        findings.append(f"{jname}.{jf}: Julia {jt} is not the mirror of {sub}")
                    if (int(mm.group(1)) if mm.group(1) else 0) != alen:
                        findings.append(f"{jname}.{jf}: array length differs from C ({alen})")
                elif not re.fullmatch(r"[A-Z]\w*", jt):
                    findings.append(f"{jname}.{jf}: expected a type parameter or a Cmx mirror for C member {sub}, got {jt}")
        # size contract
        if cname in hdr.sizes and hdr.size_of(cname) != hdr.sizes[cname]:
            findings.append(f"{cname}: flattened size {hdr.size_of(cname)} differs from CMX_ASSERT_PARAM_STRUCT_SIZES {hdr.sizes[cname]}")
    for cname in hdr.structs:
        if cname not in hdr.sizes:
            findings.append(f"{cname}: no entry in CMX_ASSERT_PARAM_STRUCT_SIZES")


def ref_leaves(ref: Reference, name: str, nested: dict, findings, where: str):
    """Flattened (translit leaf name, array length or 0 or None) of a reference struct."""
    out = []
    for f, t in ref.structs[name]:
        t = (t or "").replace(" ", "")
        if f in nested:
            sub = nested[f].split(".")[-1]
            if sub not in ref.structs:
                findings.append(f"{where}: nested type {nested[f]} of field {f} is not a reference struct")
                continue
            out += [((nm, f"{translit(f)}_{nm}") if isinstance(nm, str) else nm, a) for nm, a in ref_leaves(ref, sub, {}, findings, where)]
        elif t == "FT":
            out.append((translit(f), 0))
        elif t.startswith("NTuple{"):
            n = t[len("NTuple{"):].split(",")[0]
            out.append((translit(f), int(n) if n.isdigit() else None))
        else:
            findings.append(f"{where}: field {f}::{t} is neither FT nor listed with its concrete type in DIRECT_LAYOUT")
    return out


def check_direct(shim: Shim, hdr: Header, ref: Reference, findings):
    for refname, nested, cname in shim.direct:
        base = refname.split(".")[-1]
        where = f"DIRECT_LAYOUT {refname} => {cname}"
        if cname not in hdr.structs:
            findings.append(f"{where}: no such C struct")
            continue
        if base not in ref.structs:
            findings.append(f"{where}: the reference defines no struct {base}")
            continue
        got = ref_leaves(ref, base, nested, findings, where)
        want = [(nm, arg) for nm, kind, arg in hdr.leaves(cname)]
        same = len(got) == len(want) and all(w[0] == g[0] or (isinstance(g[0], tuple) and w[0] in g[0]) for g, w in zip(got, want))
        if not same:
            findings.append(f"{where}: field lists differ\n    reference (transliterated): {[g[0] for g in got]}\n    C:                          {[w[0] for w in want]}")
            continue
        for (gn, ga), (wn, wa) in zip(got, want):
            if ga is not None and ga != wa:
                findings.append(f"{where}: {gn}: tuple length {ga} vs C array length {wa}")


def check_accesses(shim: Shim, ref: Reference, findings):
    prefixes = {"CMP": "CMP", "AM": "AM", "QUAD": "QUAD", "BMT": "BMT", "AA": "AA"}
    n_checked = 0
    for name, args, body, line in shim.functions:
        for var, ty in args.items():
            acc = re.findall(rf"(?<![\w.]){re.escape(var)}\.({IDENT})((?:\.{IDENT})*)", body)
            if not acc:
                continue
            if ty == "NamedTuple":
                keys = ref.process_param_keys if name == "pack_process_params" else ref.process_param_inner
                for f, chain in acc:
                    n_checked += 1
                    if f not in keys:
                        findings.append(f"line {line} {name}: process_params has no entry '{f}' (reference: src/parameters/Microphysics1MOptions.jl)")
                    if chain and f != "frostenberg":
                        findings.append(f"line {line} {name}: nested access {var}.{f}{chain}")
                continue
            mod, _, base = ty.partition(".")
            if mod not in prefixes or not base:
                continue
            if base not in ref.structs:
                if not ref.defines(mod, base):
                    findings.append(f"line {line} {name}: argument type {ty} is not defined by the reference")
                continue
            have = {f for f, _ in ref.structs[base]}
            for f, chain in acc:
                n_checked += 1
                if f not in have:
                    findings.append(f"line {line} {name}: {ty} has no field '{f}' (fields: {sorted(have)})")
                if chain:
                    findings.append(f"line {line} {name}: nested access {var}.{f}{chain} — one level per method (pack the member through its own method)")
    return n_checked


def check_names(shim: Shim, ref: Reference, findings):
    n = 0
    for mod in ("CMP", "BMT", "AM", "AA", "QUAD"):
        for nm in sorted(set(re.findall(rf"\b{mod}\.({IDENT})", shim.nostr))):
            n += 1
            if not ref.defines(mod, nm):
                findings.append(f"{mod}.{nm}: not defined in {Reference.MODULE_FILES[mod]}")
    for nm in sorted(set(re.findall(rf"\bTDP\.({IDENT})", shim.nostr))):
        n += 1
        if nm not in Reference.TDP_USED_BY_REFERENCE | Reference.TDP_THERMODYNAMICS_OWN:
            findings.append(f"TDP.{nm}: not an accessor the reference or Thermodynamics.jl's parameter API is known to have")
    return n


def check_ccalls(shim: Shim, hdr: Header, findings):
    seen = set()
    for name, ret, types, args, line in shim.ccalls:
        seen.add(name)
        if name not in hdr.protos:
            findings.append(f"line {line}: ccall of {name}: no such entry family in include/cmx.h")
            continue
        if ret.strip() != "Int32":
            findings.append(f"line {line}: {name}: return type {ret}, the ABI returns int32_t")
        want = hdr.protos[name]
        if types != want:
            diff = [f"#{k + 1}: julia {a} / C {b}" for k, (a, b) in enumerate(zip(types, want)) if a != b]
            findings.append(f"line {line}: {name}: argument types differ from the prototype ({len(types)} vs {len(want)} parameters) {diff}")
        if len(args) != len(types):
            findings.append(f"line {line}: {name}: {len(args)} argument values for {len(types)} argument types")
    missing = sorted(set(hdr.protos) - seen)
    if missing:
        findings.append(f"entry families without a binding in julia/CMXExt.jl: {missing}")
    return len(seen)


def check_consts(shim: Shim, hdr: Header, findings):
    n = 0
    for nm, v in shim.consts.items():
        if nm in ("CMX_VERSION_MAJOR", "CMX_VERSION_MINOR") or nm in hdr.defines:
            n += 1
            if hdr.defines.get(nm) != v:
                findings.append(f"const {nm} = {v}, include/cmx.h says {hdr.defines.get(nm)}")
        else:
            findings.append(f"const {nm}: not defined in include/cmx.h")
    return n


def run(verbose: bool = True, shim_text: str = None):
    """All checks; `shim_text` substitutes the text of julia/CMXExt.jl (the mutation tests of tests/test_julia_shim.py)."""
    findings = []
    hdr, shim = Header(), Shim(text=shim_text)
    lint(shim, findings)
    check_mirrors(shim, hdr, findings)
    n_cc = check_ccalls(shim, hdr, findings)
    n_const = check_consts(shim, hdr, findings)
    summary = {"mirror_structs": len(shim.mirrors), "direct_layout_rows": len(shim.direct), "ccall_families": n_cc, "constants": n_const,
               "functions": len(shim.functions), "reference": REFERENCE.exists()}
    if REFERENCE.exists():
        ref = Reference()
        check_direct(shim, hdr, ref, findings)
        summary["field_accesses"] = check_accesses(shim, ref, findings)
        summary["qualified_names"] = check_names(shim, ref, findings)
    covered = {c for _, (c, _) in shim.mirrors.items()} | {c for _, _, c in shim.direct}
    uncovered = sorted(set(hdr.structs) - covered)
    summary["c_structs_without_julia_side"] = uncovered
    if uncovered:
        findings.append(f"C parameter structs with neither a mirror nor a DIRECT_LAYOUT row: {uncovered}")
    if verbose:
        print("check_julia_shim:", summary)
        for f in findings:
            print("FINDING:", f)
        print("OK" if not findings else f"{len(findings)} finding(s)")
    return findings, summary


if __name__ == "__main__":
    sys.exit(1 if run()[0] else 0)
