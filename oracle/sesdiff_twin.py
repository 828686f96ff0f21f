"""Restatement of the edit scripts analiticcl's confusables are matched against.  TEST INFRASTRUCTURE ONLY.

The reference calls `sesdiff::shortest_edit_script(input, candidate, false, false, false)` (/root/reference/src/lib.rs:1736;
`sesdiff` 0.3.1, Cargo.toml:26), which maps the chunks of `dissimilar::diff` 1:1 to Identity / Deletion / Insertion
instructions.  Neither crate is vendored in /root/reference, so this file restates their PUBLISHED algorithm:
`dissimilar` is a port of the Diff part of Google's diff-match-patch -- common prefix/suffix, containment speed-up,
Myers bisect (no half-match: that is only used with a deadline), cleanup_semantic (equality elimination, lossless
shifts, overlap extraction) and cleanup_merge -- operating on Unicode scalar values.

PARITY UNPINNED beyond the reference's own four tests (tests/main.rs:914-1020: `-[y]+[i]` on huys -> huis / huls and a
non-matching pattern): tie-breaking of the bisect and the clean-up rules follow diff-match-patch as published."""
from typing import List, Tuple

EQ, DEL, INS = "=", "-", "+"


def _common_prefix(a, b):
    n = min(len(a), len(b))
    i = 0
    while i < n and a[i] == b[i]:
        i += 1
    return i


def _common_suffix(a, b):
    n = min(len(a), len(b))
    i = 0
    while i < n and a[len(a) - 1 - i] == b[len(b) - 1 - i]:
        i += 1
    return i


def _common_overlap(a, b):
    """length of the longest suffix of a that is a prefix of b"""
    n = min(len(a), len(b))
    for k in range(n, 0, -1):
        if a[len(a) - k:] == b[:k]:
            return k
    return 0


def _main(a: str, b: str) -> List[Tuple[str, str]]:
    if a == b:
        return [(EQ, a)] if a else []
    p = _common_prefix(a, b)
    prefix, a, b = a[:p], a[p:], b[p:]
    s = _common_suffix(a, b)
    suffix = a[len(a) - s:] if s else ""
    if s:
        a, b = a[:len(a) - s], b[:len(b) - s]
    diffs = _compute(a, b)
    if prefix:
        diffs.insert(0, (EQ, prefix))
    if suffix:
        diffs.append((EQ, suffix))
    _cleanup_merge(diffs)
    return diffs


def _compute(a: str, b: str) -> List[Tuple[str, str]]:
    if not a:
        return [(INS, b)] if b else []
    if not b:
        return [(DEL, a)]
    longt, short = (a, b) if len(a) > len(b) else (b, a)
    i = longt.find(short)
    if i != -1:
        op = DEL if len(a) > len(b) else INS
        out = []
        if longt[:i]:
            out.append((op, longt[:i]))
        out.append((EQ, short))
        if longt[i + len(short):]:
            out.append((op, longt[i + len(short):]))
        return out
    if len(short) == 1:
        return [(DEL, a), (INS, b)]
    return _bisect(a, b)


def _bisect(a: str, b: str) -> List[Tuple[str, str]]:
    n1, n2 = len(a), len(b)
    max_d = (n1 + n2 + 1) // 2
    v_offset, v_length = max_d, 2 * max_d
    v1, v2 = [-1] * v_length, [-1] * v_length
    v1[v_offset + 1] = 0
    v2[v_offset + 1] = 0
    delta = n1 - n2
    front = delta % 2 != 0
    k1start = k1end = k2start = k2end = 0
    for d in range(max_d):
        for k1 in range(-d + k1start, d + 1 - k1end, 2):
            k1_offset = v_offset + k1
            if k1 == -d or (k1 != d and v1[k1_offset - 1] < v1[k1_offset + 1]):
                x1 = v1[k1_offset + 1]
            else:
                x1 = v1[k1_offset - 1] + 1
            y1 = x1 - k1
            while x1 < n1 and y1 < n2 and a[x1] == b[y1]:
                x1 += 1
                y1 += 1
            v1[k1_offset] = x1
            if x1 > n1:
                k1end += 2
            elif y1 > n2:
                k1start += 2
            elif front:
                k2_offset = v_offset + delta - k1
                if 0 <= k2_offset < v_length and v2[k2_offset] != -1:
                    x2 = n1 - v2[k2_offset]
                    if x1 >= x2:
                        return _main(a[:x1], b[:y1]) + _main(a[x1:], b[y1:])
        for k2 in range(-d + k2start, d + 1 - k2end, 2):
            k2_offset = v_offset + k2
            if k2 == -d or (k2 != d and v2[k2_offset - 1] < v2[k2_offset + 1]):
                x2 = v2[k2_offset + 1]
            else:
                x2 = v2[k2_offset - 1] + 1
            y2 = x2 - k2
            while x2 < n1 and y2 < n2 and a[n1 - x2 - 1] == b[n2 - y2 - 1]:
                x2 += 1
                y2 += 1
            v2[k2_offset] = x2
            if x2 > n1:
                k2end += 2
            elif y2 > n2:
                k2start += 2
            elif not front:
                k1_offset = v_offset + delta - k2
                if 0 <= k1_offset < v_length and v1[k1_offset] != -1:
                    x1 = v1[k1_offset]
                    y1 = v_offset + x1 - k1_offset
                    x2m = n1 - x2
                    if x1 >= x2m:
                        return _main(a[:x1], b[:y1]) + _main(a[x1:], b[y1:])
    return [(DEL, a), (INS, b)]


def _cleanup_merge(diffs: List[Tuple[str, str]]) -> None:
    diffs.append((EQ, ""))
    pointer = 0
    count_delete = count_insert = 0
    text_delete = text_insert = ""
    while pointer < len(diffs):
        op, text = diffs[pointer]
        if op == INS:
            count_insert += 1
            text_insert += text
            pointer += 1
        elif op == DEL:
            count_delete += 1
            text_delete += text
            pointer += 1
        else:
            if count_delete + count_insert > 1:
                if count_delete != 0 and count_insert != 0:
                    cl = _common_prefix(text_insert, text_delete)
                    if cl:
                        x = pointer - count_delete - count_insert - 1
                        if x >= 0 and diffs[x][0] == EQ:
                            diffs[x] = (EQ, diffs[x][1] + text_insert[:cl])
                        else:
                            diffs.insert(0, (EQ, text_insert[:cl]))
                            pointer += 1
                        text_insert, text_delete = text_insert[cl:], text_delete[cl:]
                    cl = _common_suffix(text_insert, text_delete)
                    if cl:
                        diffs[pointer] = (EQ, text_insert[len(text_insert) - cl:] + diffs[pointer][1])
                        text_insert, text_delete = text_insert[:len(text_insert) - cl], text_delete[:len(text_delete) - cl]
                new_ops = []
                if text_delete:
                    new_ops.append((DEL, text_delete))
                if text_insert:
                    new_ops.append((INS, text_insert))
                pointer -= count_delete + count_insert
                diffs[pointer:pointer + count_delete + count_insert] = new_ops
                pointer += len(new_ops) + 1
            elif pointer != 0 and diffs[pointer - 1][0] == EQ:
                diffs[pointer - 1] = (EQ, diffs[pointer - 1][1] + diffs[pointer][1])
                del diffs[pointer]
            else:
                pointer += 1
            count_insert = count_delete = 0
            text_delete = text_insert = ""
    if diffs[-1][1] == "":
        diffs.pop()
    changes = False
    pointer = 1
    while pointer < len(diffs) - 1:
        if diffs[pointer - 1][0] == EQ and diffs[pointer + 1][0] == EQ:
            prev_t, cur_t, next_t = diffs[pointer - 1][1], diffs[pointer][1], diffs[pointer + 1][1]
            if prev_t and cur_t.endswith(prev_t):
                diffs[pointer] = (diffs[pointer][0], prev_t + cur_t[:len(cur_t) - len(prev_t)])
                diffs[pointer + 1] = (EQ, prev_t + next_t)
                del diffs[pointer - 1]
                changes = True
            elif next_t and cur_t.startswith(next_t):
                diffs[pointer - 1] = (EQ, prev_t + next_t)
                diffs[pointer] = (diffs[pointer][0], cur_t[len(next_t):] + next_t)
                del diffs[pointer + 1]
                changes = True
        pointer += 1
    if changes:
        _cleanup_merge(diffs)


def _semantic_score(one: str, two: str) -> int:
    if not one or not two:
        return 6
    c1, c2 = one[-1], two[0]
    na1, na2 = not c1.isalnum(), not c2.isalnum()
    ws1, ws2 = na1 and c1.isspace(), na2 and c2.isspace()
    lb1, lb2 = ws1 and c1 in "\r\n", ws2 and c2 in "\r\n"
    bl1 = lb1 and (one.endswith("\n\n") or one.endswith("\n\r\n"))
    bl2 = lb2 and (two.startswith("\n\n") or two.startswith("\r\n\n") or two.startswith("\n\r\n") or two.startswith("\r\n\r\n"))
    if bl1 or bl2:
        return 5
    if lb1 or lb2:
        return 4
    if na1 and not ws1 and ws2:
        return 3
    if ws1 or ws2:
        return 2
    if na1 or na2:
        return 1
    return 0


def _cleanup_semantic_lossless(diffs: List[Tuple[str, str]]) -> None:
    pointer = 1
    while pointer < len(diffs) - 1:
        if diffs[pointer - 1][0] == EQ and diffs[pointer + 1][0] == EQ:
            eq1, edit, eq2 = diffs[pointer - 1][1], diffs[pointer][1], diffs[pointer + 1][1]
            co = _common_suffix(eq1, edit)
            if co:
                cs = edit[len(edit) - co:]
                eq1 = eq1[:len(eq1) - co]
                edit = cs + edit[:len(edit) - co]
                eq2 = cs + eq2
            best = (eq1, edit, eq2)
            best_score = _semantic_score(eq1, edit) + _semantic_score(edit, eq2)
            while edit and eq2 and edit[0] == eq2[0]:
                eq1 += edit[0]
                edit = edit[1:] + eq2[0]
                eq2 = eq2[1:]
                sc = _semantic_score(eq1, edit) + _semantic_score(edit, eq2)
                if sc >= best_score:
                    best_score, best = sc, (eq1, edit, eq2)
            if diffs[pointer - 1][1] != best[0]:
                if best[0]:
                    diffs[pointer - 1] = (EQ, best[0])
                else:
                    del diffs[pointer - 1]
                    pointer -= 1
                diffs[pointer] = (diffs[pointer][0], best[1])
                if best[2]:
                    diffs[pointer + 1] = (EQ, best[2])
                else:
                    del diffs[pointer + 1]
                    pointer -= 1
        pointer += 1


def _cleanup_semantic(diffs: List[Tuple[str, str]]) -> None:
    changes = False
    equalities: List[int] = []
    last_eq = None
    pointer = 0
    li1 = ld1 = li2 = ld2 = 0
    while pointer < len(diffs):
        if diffs[pointer][0] == EQ:
            equalities.append(pointer)
            li1, li2, ld1, ld2 = li2, 0, ld2, 0
            last_eq = diffs[pointer][1]
        else:
            if diffs[pointer][0] == INS:
                li2 += len(diffs[pointer][1])
            else:
                ld2 += len(diffs[pointer][1])
            if last_eq and len(last_eq) <= max(li1, ld1) and len(last_eq) <= max(li2, ld2):
                diffs.insert(equalities[-1], (DEL, last_eq))
                diffs[equalities[-1] + 1] = (INS, diffs[equalities[-1] + 1][1])
                equalities.pop()
                if equalities:
                    equalities.pop()
                pointer = equalities[-1] if equalities else -1
                li1 = ld1 = li2 = ld2 = 0
                last_eq = None
                changes = True
        pointer += 1
    if changes:
        _cleanup_merge(diffs)
    _cleanup_semantic_lossless(diffs)
    pointer = 1
    while pointer < len(diffs):
        if diffs[pointer - 1][0] == DEL and diffs[pointer][0] == INS:
            deletion, insertion = diffs[pointer - 1][1], diffs[pointer][1]
            o1 = _common_overlap(deletion, insertion)
            o2 = _common_overlap(insertion, deletion)
            if o1 >= o2:
                if o1 >= len(deletion) / 2.0 or o1 >= len(insertion) / 2.0:
                    diffs.insert(pointer, (EQ, insertion[:o1]))
                    diffs[pointer - 1] = (DEL, deletion[:len(deletion) - o1])
                    diffs[pointer + 1] = (INS, insertion[o1:])
                    pointer += 1
            else:
                if o2 >= len(deletion) / 2.0 or o2 >= len(insertion) / 2.0:
                    diffs.insert(pointer, (EQ, deletion[:o2]))
                    diffs[pointer - 1] = (INS, insertion[:len(insertion) - o2])
                    diffs[pointer + 1] = (DEL, deletion[o2:])
                    pointer += 1
            pointer += 1
        pointer += 1


def shortest_edit_script(source: str, target: str) -> List[Tuple[str, str]]:
    """[(op, text)] with op in '=', '-', '+' (sesdiff Identity / Deletion / Insertion)."""
    diffs = _main(source, target)
    _cleanup_semantic(diffs)
    _cleanup_merge(diffs)
    return [(op, t) for op, t in diffs if t != ""]


def script_to_str(script) -> str:
    return "".join(f"{op}[{t}]" for op, t in script)


class Confusable:
    """/root/reference/src/confusables.rs:5-128"""

    def __init__(self, editscript: str, weight: float):
        self.strictbegin = editscript[0:1] == "^"
        self.strictend = editscript[-1:] == "$"
        body = editscript[1 if self.strictbegin else 0: len(editscript) - (1 if self.strictend else 0)]
        self.instructions = []  # (op, [options])
        begin = 0
        for i, c in enumerate(body):
            if c == "]":
                ins = body[begin:i + 1]
                if len(ins) <= 3 or ins[1] != "[" or ins[0] not in "=+-":
                    raise ValueError(f"invalid edit instruction {ins!r}")
                self.instructions.append((ins[0], ins[2:-1].split("|")))
                begin = i + 1
        self.weight = weight

    def found_in(self, refscript) -> bool:
        l = len(self.instructions)
        matches = 0
        for i, (rop, sref) in enumerate(refscript):
            if matches < l:
                op, opts = self.instructions[matches]
                found = False
                if op == rop:
                    for s in opts:
                        if op in "+-":
                            ok = sref.endswith(s)
                        elif matches == 0 and matches == l - 1:
                            ok = s == sref
                        elif matches == 0:
                            ok = sref.endswith(s)
                        elif matches == l - 1:
                            ok = sref.startswith(s)
                        else:
                            ok = s == sref
                        if ok:
                            found = True
                            break
                if not found:
                    matches = 0
                    if self.strictbegin:
                        return False
                    continue
                matches += 1
                if matches == l:
                    return i == len(refscript) - 1 if self.strictend else True
        return False
