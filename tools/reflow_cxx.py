#!/usr/bin/env python3
"""tools/reflow_cxx.py [--check] [--limit N] FILE...: break the over-long lines of C++ / HIP sources (there is no clang-format in this image).

A line longer than the limit is re-emitted one statement per line: it is cut after every ';' and '{' and before every '}' that lies outside parentheses, strings and
comments (a brace group without a statement in it -- an initialiser list -- stays whole), nested blocks are indented, a trailing `// comment` moves to its own line(s) above, and
a piece that is still too long is wrapped behind a ", " / " && " / " || " / " ? " / " : " / " = " / " + " (a "<<" for stream chains) outside string literals, or in front of its deepest '(' when no such point exists.
Preprocessor lines, macro bodies (lines ending in a backslash or following one) and lines with an unterminated string are left alone.  Comment-only lines are re-wrapped at word
boundaries.

Only white space and the position of comments change: the file is rewritten only when its token stream -- the text with comments and all white space outside
string / character literals removed -- is identical before and after (the same rule tools/reflow.py applies to Python with the AST)."""
import re, sys

def scan(line):
    """-> (code, comment): the line cut at the `//` that starts a trailing comment (None when there is none or the line has an unterminated literal / a block comment)"""
    i, n, q = 0, len(line), None
    while i < n:
        c = line[i]
        if q:
            if c == "\\": i += 2; continue
            if c == q: q = None
        elif c in "\"'":
            if c == "'" and i > 0 and (line[i - 1].isalnum()): pass      # digit separator 1'000 (not used here, but harmless)
            else: q = c
        elif c == "/" and i + 1 < n and line[i + 1] == "/": return line[:i].rstrip(), line[i:]
        elif c == "/" and i + 1 < n and line[i + 1] == "*": return None, None
        i += 1
    return (None, None) if q else (line.rstrip(), "")

def tokens_of(text):
    """the token stream as a string: comments removed, white space kept (as one blank) only between two identifier characters; blanks inside literals are protected"""
    out, i, n, q = [], 0, len(text), None
    while i < n:
        c = text[i]
        if q:
            out.append("\x00" if c == " " else c)
            if c == "\\": out.append(text[i + 1]); i += 2; continue
            if c == q: q = None
        elif c in "\"'": q = c; out.append(c)
        elif c == "/" and text[i + 1:i + 2] == "/":
            while i < n and text[i] != "\n": i += 1
            continue
        elif c == "/" and text[i + 1:i + 2] == "*":
            j = text.find("*/", i + 2); i = n if j < 0 else j + 2; out.append(" "); continue
        elif c.isspace():
            if c == "\n" and out and out[-1] == "\\": out.pop()      # line continuation
            if out and out[-1] != " ": out.append(" ")
        else: out.append(c)
        i += 1
    return re.sub(r"(?<![A-Za-z0-9_]) +| +(?![A-Za-z0-9_])", "", "".join(out))

def pieces(code):
    """cut at ';' '{' '}' outside parentheses / brackets / literals: [(text, kind)], kind in 'open' (ends with '{'), 'close' (starts with '}'), 'stmt'"""
    out, cur, depth, q, i, n = [], [], 0, None, 0, len(code)
    def brace_is_block(j):      # does the group opened at j hold a ';' at its own level (a block) -- or nothing like that (an initialiser list)?
        d, p, k, qq = 0, 0, j + 1, None
        while k < n:
            ch = code[k]
            if qq:
                if ch == "\\": k += 2; continue
                if ch == qq: qq = None
            elif ch in "\"'": qq = ch
            elif ch in "([": p += 1
            elif ch in ")]": p -= 1
            elif ch == "{": d += 1
            elif ch == "}":
                if d == 0: return False
                d -= 1
            elif ch == ";" and p == 0: return True
            k += 1
        return True      # the block goes on past this line
    stack = []      # per open brace: is it a block we cut at?
    while i < n:
        c = code[i]
        if q:
            cur.append(c)
            if c == "\\": cur.append(code[i + 1]); i += 2; continue
            if c == q: q = None
        elif c in "\"'": q = c; cur.append(c)
        elif c in "([": depth += 1; cur.append(c)
        elif c in ")]": depth -= 1; cur.append(c)
        elif c == "{":
            blk = depth == 0 and brace_is_block(i) and not any(s is False for s in stack)
            stack.append(blk if depth == 0 else None); cur.append(c)
            if blk: out.append(("".join(cur).strip(), "open")); cur = []
        elif c == "}":
            blk = stack.pop() if stack else (depth == 0)
            if blk:
                if "".join(cur).strip(): out.append(("".join(cur).strip(), "stmt"))
                cur = ["}"]
                # what follows the brace on the same piece: `} else {`, `};`, `} while (..);`
                j = i + 1
                while j < n and code[j] == " ": j += 1
                if code[j:j + 1] in (";", ","): cur.append(code[j]); i = j
                out.append(("".join(cur), "close")); cur = []
            else: cur.append(c)
        elif c == ";" and depth == 0 and not any(s is False for s in stack):
            cur.append(c); out.append(("".join(cur).strip(), "stmt")); cur = []
        else: cur.append(c)
        i += 1
    if "".join(cur).strip(): out.append(("".join(cur).strip(), "stmt"))
    # glue `}` + `else ... {` / `while (...);` back together
    glued = []
    for t, k in out:
        if glued and glued[-1][1] == "close" and glued[-1][0] == "}" and re.match(r"(else\b|while\b|catch\b)", t): glued[-1] = ("} " + t, "close_open" if k == "open" else "close")
        else: glued.append((t, k))
    return glued

def wrap(text, indent, limit):
    """break one statement outside literals: at the shallowest nesting level, behind a comma if there is one, else at a logical operator, else at another operator; continuation
    lines are indented by 4 more"""
    lines, pad = [], " " * indent
    while len(pad + text) > limit:
        room = limit - len(pad); q, depth, i, cands = None, 0, 0, []
        while i < len(text) and i < room:
            c = text[i]
            if q:
                if c == "\\": i += 2; continue
                if c == q: q = None
            elif c in "\"'": q = c
            elif c in "([{": depth += 1
            elif c in ")]}": depth -= 1
            elif c == " " and i > 8:
                prev, nxt = text[:i], text[i + 1:i + 3]
                if prev.endswith(","): cands.append((depth, 0, i))
                elif prev.endswith(("&&", "||")) or nxt in ("&&", "||"): cands.append((depth, 1, i))
                elif prev.endswith((" ?", " :", " =", "<<", "+=", "|=", "-=")) or nxt in ("<<", "? ", ": "): cands.append((depth, 2, i))
                elif prev.endswith((" +", " -")): cands.append((depth, 3, i))
            i += 1
        cands = [c for c in cands if c[2] >= room // 4] or cands
        if not cands: break
        best = max(cands, key=lambda c: (-c[0], -c[1], c[2]))[2]
        lines.append(pad + text[:best].rstrip()); text = text[best + 1:].lstrip(); pad = " " * (indent + 4)
    lines.append(pad + text)
    return lines

def wrap_comment(indent, comment, limit):
    body = comment[2:].strip(); pad = " " * indent + "// "; words = body.split(" "); lines, cur = [], ""
    for w in words:
        if cur and len(pad) + len(cur) + 1 + len(w) > limit: lines.append(pad + cur); cur = w
        else: cur = (cur + " " + w) if cur else w
    if cur: lines.append(pad + cur)
    return lines

def reflow(src, limit, target):
    out, in_macro = [], False
    for line in src.split("\n"):
        stripped = line.lstrip(); was_macro = in_macro
        in_macro = line.rstrip().endswith("\\")
        if len(line) <= limit or was_macro or in_macro or stripped.startswith("#"): out.append(line); continue
        indent = len(line) - len(stripped)
        if stripped.startswith("//"):
            if stripped.startswith("///") or "  " in stripped[3:].strip(): out.append(line)      # (tables / aligned text: leave)
            else: out.extend(wrap_comment(indent, stripped, target))
            continue
        code, comment = scan(line)
        if code is None: out.append(line); continue
        if comment: out.extend(wrap_comment(indent, comment, target))
        level = 0
        for text, kind in pieces(code[indent:]):
            if kind in ("close", "close_open"): level = max(0, level - 1)
            out.extend(wrap(text, indent + 2 * level, target))
            if kind in ("open", "close_open"): level += 1
    return "\n".join(out)

def main():
    args = sys.argv[1:]; check = "--check" in args; args = [a for a in args if a != "--check"]
    limit, target = 180, 160
    if "--limit" in args: k = args.index("--limit"); limit = int(args[k + 1]); del args[k:k + 2]
    bad = 0
    for path in args:
        src = open(path).read(); new = reflow(src, limit, target)
        if tokens_of(new) != tokens_of(src): print(f"{path}: token stream would change -- left alone"); bad += 1; continue
        longest = max(len(l) for l in new.split("\n")); n_over = sum(len(l) > 200 for l in new.split("\n"))
        print(f"{path}: {src.count(chr(10)) + 1} -> {new.count(chr(10)) + 1} lines, longest {longest}, over 200: {sum(len(l) > 200 for l in src.split(chr(10)))} -> {n_over}")
        if not check and new != src: open(path, "w").write(new)
    sys.exit(1 if bad else 0)

if __name__ == "__main__": main()
