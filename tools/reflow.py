#!/usr/bin/env python3
"""Re-flow a dense Python source file without changing what it does: one statement per line (``a; b`` and
``if x: a; b`` become separate lines / blocks) and bracketed expressions wrapped at their commas, long string
literals split into adjacent literals, so that no line is wider than --width where the grammar allows it.

The result is accepted only if its AST equals the input's (``ast.dump`` without positions), i.e. the change is
whitespace, line breaks and the spelling of adjacent string literals -- nothing a reviewer has to re-verify.

    python tools/reflow.py bench.py [--width 160] [--check]
"""
import ast, io, sys, tokenize, argparse, re

COMPOUND = {"if", "elif", "else", "for", "while", "with", "def", "class", "try", "except", "finally", "async"}
OPEN, CLOSE = "([{", ")]}"


def logical_lines(src):
    """[(tokens of one logical line incl. its NEWLINE, leading non-code tokens (comments / NL / INDENT / DEDENT))]"""
    toks = list(tokenize.generate_tokens(io.StringIO(src).readline))
    out, cur = [], []
    for t in toks:
        cur.append(t)
        if t.type in (tokenize.NEWLINE, tokenize.ENDMARKER):
            out.append(cur); cur = []
    if cur:
        out.append(cur)
    return out


def split_string(tok_s, room):
    """a long (f-)string literal -> adjacent literals of at most ~room characters (split at spaces outside {})"""
    m = re.match(r"^([rbfuRBFU]*)(\"\"\"|'''|\"|')", tok_s)
    if not m or len(m.group(2)) == 3 or len(tok_s) <= room:
        return [tok_s]
    pre, q = m.group(1), m.group(2)
    body = tok_s[len(pre) + 1:-1]
    isf = "f" in pre.lower()
    parts, start, depth, i, last_space = [], 0, 0, 0, -1
    while i < len(body):
        c = body[i]
        if c == "\\":
            i += 2; continue
        if isf and c == "{":
            if body[i:i + 2] == "{{": i += 2; continue
            depth += 1
        elif isf and c == "}":
            if body[i:i + 2] == "}}" and depth == 0: i += 2; continue
            depth = max(0, depth - 1)
        elif c == " " and depth == 0:
            last_space = i
        if i - start >= room and last_space > start:
            parts.append(body[start:last_space + 1]); start = last_space + 1; last_space = -1
        i += 1
    parts.append(body[start:])
    return [pre + q + p + q for p in parts if p != "" or len(parts) == 1]


def render(tokens, indent, width):
    """tokens of ONE simple statement or compound header -> lines; wraps inside brackets only"""
    lines, cur, depth = [], indent, 0
    cont = indent + "    "
    prev = None
    stack = []      # continuation indents of the open brackets

    def gap(a, b):      # the original spacing between two tokens (a line break inside brackets becomes one space, none after an opening bracket)
        if a is None: return ""
        if a.end[0] == b.start[0]: return " " * (b.start[1] - a.end[1])
        return "" if (a.string in OPEN or b.string in CLOSE) else " "

    last_break = -1      # index into `cur` right behind the last comma at bracket depth >= 1 (the preferred place to wrap)
    for idx, t in enumerate(tokens):
        s = t.string
        pieces = [s]
        if t.type == tokenize.STRING and depth > 0 and len(s) > width - len(cont) - 20:
            pieces = split_string(s, max(60, width - len(cont) - 30))
        for pi, piece in enumerate(pieces):
            sp = gap(prev, t) if pi == 0 else " "
            if depth > 0 and len(cur) + len(sp) + len(piece) > width and cur.strip():
                ci = stack[-1] if stack else cont
                if last_break > len(ci) and pi == 0:      # wrap behind the last comma, carry the rest over
                    lines.append(cur[:last_break].rstrip()); cur = ci + cur[last_break:].lstrip(); last_break = -1
                    if len(cur) + len(sp) + len(piece) > width and cur.strip():
                        lines.append(cur.rstrip()); cur = ci; sp = ""
                else:
                    lines.append(cur.rstrip()); cur = ci; sp = ""; last_break = -1
            cur += sp + piece
            if len(pieces) > 1: last_break = len(cur)
        if s in OPEN and t.type == tokenize.OP:
            depth += 1; stack.append(indent + "    " * min(depth, 3))
        elif s in CLOSE and t.type == tokenize.OP:
            depth -= 1; stack.pop()
        elif s == "," and t.type == tokenize.OP and depth > 0:
            last_break = len(cur)
        prev = t
    lines.append(cur.rstrip())
    return lines


def wrap_comment(line, width):
    """a comment-only line wider than `width` -> several comment lines"""
    if len(line) <= width: return [line]
    indent = re.match(r"\s*", line).group(0); words = line.strip()[1:].strip().split(" "); out, cur = [], indent + "#"
    for w in words:
        if len(cur) + 1 + len(w) > width and cur.strip() != "#":
            out.append(cur); cur = indent + "#"
        cur += " " + w
    out.append(cur)
    return out


def reflow(src, width):
    out_lines = []
    src_lines = src.splitlines()
    for ll in logical_lines(src):
        code = [t for t in ll if t.type not in (tokenize.NL, tokenize.COMMENT, tokenize.INDENT, tokenize.DEDENT, tokenize.NEWLINE, tokenize.ENDMARKER)]
        if not code:
            # blank / comment-only lines: copy verbatim
            rows = sorted({t.start[0] for t in ll if t.type in (tokenize.COMMENT, tokenize.NL)})
            for r in rows:
                out_lines.extend(wrap_comment(src_lines[r - 1].rstrip(), width))
            continue
        first_row, last_row = code[0].start[0], code[-1].end[0]
        lead_rows = sorted({t.start[0] for t in ll if t.type in (tokenize.COMMENT, tokenize.NL) and t.start[0] < first_row})
        for r in lead_rows:
            out_lines.extend(wrap_comment(src_lines[r - 1].rstrip(), width))
        text_rows = src_lines[first_row - 1:last_row]
        has_semicolon = any(t.type == tokenize.OP and t.string == ";" for t in code)
        too_long = any(len(r) > width for r in text_rows)      # (a trailing comment counts: it moves to its own line above)
        multi_string = any(t.type == tokenize.STRING and t.start[0] != t.end[0] for t in code)
        inline_comments = [t for t in ll if t.type == tokenize.COMMENT and first_row <= t.start[0] <= last_row]
        if multi_string or not (has_semicolon or too_long) or (len(inline_comments) > 1):
            out_lines.extend(r.rstrip() for r in text_rows)
            continue
        indent = re.match(r"\s*", src_lines[first_row - 1]).group(0)
        # split at depth-0 ';' and at the ':' that ends a compound header with an inline body
        stmts, cur, depth, lambdas = [], [], 0, 0
        header = None
        is_compound = code[0].type == tokenize.NAME and code[0].string in COMPOUND
        for t in code:
            if t.type == tokenize.OP and t.string in OPEN: depth += 1
            elif t.type == tokenize.OP and t.string in CLOSE: depth -= 1
            if t.type == tokenize.NAME and t.string == "lambda" and depth == 0: lambdas += 1
            if t.type == tokenize.OP and t.string == ":" and depth == 0:
                if lambdas: lambdas -= 1
                elif is_compound and header is None:
                    cur.append(t); header = cur; cur = []; continue
            if t.type == tokenize.OP and t.string == ";" and depth == 0:
                stmts.append(cur); cur = []; continue
            cur.append(t)
        if cur: stmts.append(cur)
        comment = ("      " + inline_comments[0].string) if inline_comments else ""
        body_indent = indent
        emitted = []
        if header is not None:
            emitted.extend(render(header, indent, width))
            body_indent = indent + "    "
            if stmts and len(stmts) == 1 and not comment and len(emitted) == 1:
                one = render(stmts[0], "", width)
                if len(one) == 1 and len(emitted[0]) + 1 + len(one[0]) <= min(width, 120):
                    emitted[0] += " " + one[0]; stmts = []
        for s_ in stmts:
            if s_: emitted.extend(render(s_, body_indent, width))
        if comment:
            if len(emitted[0]) + len(comment) <= width: emitted[0] += comment
            else: emitted[0:0] = wrap_comment(indent + inline_comments[0].string, width)
        out_lines.extend(emitted)
    return "\n".join(out_lines) + "\n"


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("path"); ap.add_argument("--width", type=int, default=160); ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    src = open(a.path).read()
    new = reflow(src, a.width)
    if ast.dump(ast.parse(src)) != ast.dump(ast.parse(new)):
        open(a.path + ".reflow_rejected", "w").write(new)
        sys.exit("AST changed: result NOT written (see %s.reflow_rejected)" % a.path)
    longest = max(len(l) for l in new.splitlines())
    print("%s: %d -> %d lines, longest line %d -> %d, lines > %d: %d -> %d" % (a.path, len(src.splitlines()), len(new.splitlines()), max(len(l) for l in src.splitlines()), longest, a.width,
          sum(len(l) > a.width for l in src.splitlines()), sum(len(l) > a.width for l in new.splitlines())))
    if not a.check: open(a.path, "w").write(new)


if __name__ == "__main__":
    main()
