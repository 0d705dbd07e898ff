"""A small C++ class-declaration reader, enough for the reference's headers (include/ORB_SLAM2/*.h) and the stand-in classes of
tests/cpp/test_dropin.cpp: per class its bases, and per member the kind (method / field / type), access (public / protected / private),
static-ness, and for methods the parameter count range (defaults counted).  Not a C++ parser: no templates of classes, no macros that
open scopes, no nested classes' members (nested classes are recorded as types).  Used by tests/test_reference_boundary.py."""
from __future__ import annotations

import re
from dataclasses import dataclass, field


def strip_comments(src: str) -> str:
    src = re.sub(r"/\*.*?\*/", lambda m: " " * 0 + "\n" * m.group(0).count("\n"), src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return re.sub(r'"(?:\\.|[^"\\])*"', '""', src)


@dataclass
class Member:
    name: str
    kind: str                 # "method" | "field" | "type"
    access: str
    static: bool = False
    arity: tuple = (0, 0)     # methods: (min, max) parameters
    decl: str = ""            # the declaration text (one line)


@dataclass
class ClassDecl:
    name: str
    bases: list = field(default_factory=list)
    members: dict = field(default_factory=dict)   # name -> [Member, ...] (overloads)
    friends: list = field(default_factory=list)


def _match_brace(s: str, i: int, open_c="{", close_c="}") -> int:
    depth = 0
    for j in range(i, len(s)):
        if s[j] == open_c:
            depth += 1
        elif s[j] == close_c:
            depth -= 1
            if depth == 0:
                return j
    raise ValueError("unbalanced braces")


def split_top(s: str, sep: str = ",") -> list:
    out, depth, cur = [], 0, ""
    s = s.replace("->", "\x01\x02")   # not a closing angle bracket
    for ch in s:
        if ch in "(<[{":
            depth += 1
        elif ch in ")>]}":
            depth -= 1
        if ch == sep and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur)
    return [x.replace("\x01\x02", "->") for x in out]


def _statements(body: str):
    """top-level statements of a class body: text up to ';' or a '{...}' block (function bodies are dropped, an initialiser list too)"""
    i, n, cur = 0, len(body), ""
    while i < n:
        ch = body[i]
        if ch == "{":
            j = _match_brace(body, i)
            stmt = cur.strip()
            # `= {..};` / `{...};` initialisers of fields continue to the ';', function bodies end the statement
            k = j + 1
            while k < n and body[k] in " \t\r\n":
                k += 1
            if k < n and body[k] == ";" and "(" not in stmt.split("=")[0].split(":")[-1][-200:] and not re.search(r"\)\s*(const)?\s*(override)?\s*(noexcept)?\s*$", stmt):
                yield stmt
                i, cur = k + 1, ""
                continue
            yield stmt
            cur = ""
            i = j + 1
            if i < n and body[i] == ";":
                i += 1
            continue
        if ch == ";":
            yield cur.strip()
            cur = ""
        else:
            cur += ch
        i += 1
    if cur.strip():
        yield cur.strip()


_ACCESS = re.compile(r"^\s*(public|protected|private)\s*:\s*")


def parse_classes(src: str) -> dict:
    src = strip_comments(src)
    out = {}
    for m in re.finditer(r"\b(class|struct)\s+([A-Za-z_]\w*)\s*(:[^{;]*)?\{", src):
        kind, name, bases = m.group(1), m.group(2), m.group(3) or ""
        start = m.end() - 1
        try:
            end = _match_brace(src, start)
        except ValueError:
            continue
        body = src[start + 1:end]
        cd = ClassDecl(name, [re.sub(r"\b(public|protected|private|virtual)\b", "", b).strip().split("::")[-1]
                              for b in split_top(bases.lstrip(":")) if b.strip()])
        access = "public" if kind == "struct" else "private"
        for st in _statements(body):
            while True:
                am = _ACCESS.match(st)
                if not am:
                    break
                access = am.group(1)
                st = st[am.end():]
            st = " ".join(st.split())
            if not st:
                continue
            if st.startswith("friend "):
                cd.friends.append(st[7:].replace("class ", "").strip())
                continue
            if re.match(r"^(typedef|using)\b", st):
                tm = re.search(r"(\w+)\s*(=.*)?$", st if st.startswith("using") else st)
                nm = re.match(r"using\s+(\w+)\s*=", st)
                tname = nm.group(1) if nm else re.search(r"(\w+)$", st).group(1)
                cd.members.setdefault(tname, []).append(Member(tname, "type", access, decl=st))
                continue
            if re.match(r"^(class|struct|enum)\b", st) and "(" not in st:
                nm = re.match(r"^(?:class|struct|enum(?:\s+class)?)\s+(\w+)", st)
                if nm:
                    cd.members.setdefault(nm.group(1), []).append(Member(nm.group(1), "type", access, decl=st))
                continue
            # constructor initialiser lists: cut at the ':' that follows the parameter list
            static = bool(re.match(r"^(inline\s+)?static\b|^static\b", st)) or " static " in " " + st.split("(")[0] + " "
            head = st
            pm = re.search(r"([~\w]+|operator\s*[^\s(]+)\s*\(", head)
            is_method = False
            if pm and not re.search(r"=\s*[^=(]*$", head.split("(")[0]):
                # a '(' that belongs to a declarator, not to an initialiser `T x = f(...)`
                pre = head[:pm.start()]
                if "=" not in pre:
                    is_method = True
            if is_method:
                name_m = pm.group(1).replace(" ", "")
                po = head.index("(", pm.start())
                pc = _match_brace(head, po, "(", ")")
                params = [p for p in split_top(head[po + 1:pc]) if p.strip() and p.strip() != "void"]
                n_def = sum(1 for p in params if "=" in p)
                cd.members.setdefault(name_m, []).append(Member(name_m, "method", access, static, (len(params) - n_def, len(params)), st))
            else:
                decl = re.split(r"=|\{", st)[0].strip()
                names = split_top(decl)
                first = names[0]
                fm = re.search(r"([A-Za-z_]\w*)\s*(\[[^\]]*\])?$", first)
                if not fm:
                    continue
                tpart = first[:fm.start()].strip()
                all_names = [fm.group(1)] + [re.search(r"([A-Za-z_]\w*)", x).group(1) for x in names[1:] if re.search(r"[A-Za-z_]\w*", x)]
                for nm_ in all_names:
                    cd.members.setdefault(nm_, []).append(Member(nm_, "field", access, static, decl=tpart + " " + nm_))
        out[name] = cd
    return out


def resolve(classes: dict, name: str, seen=None) -> dict:
    """members of a class including its bases (derived first)"""
    seen = seen or set()
    if name not in classes or name in seen:
        return {}
    seen.add(name)
    merged = {}
    for b in classes[name].bases:
        merged.update(resolve(classes, b, seen))
    merged.update(classes[name].members)
    return merged
