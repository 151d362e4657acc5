"""LaTeX math -> SymPy, as a hand-written tokenizer + recursive-descent parser.

The reference vendors an ANTLR-generated parser for the same job (``evaluation/latex2sympy/latex2sympy2.py``,
grammar ``PS.g4``) and grades answers through it (``evaluation/grader.py:281-335``). ANTLR's runtime is not a
dependency here: the grammar a math grader needs is small enough to parse by hand.

Covered: integers / decimals / percent; Latin and Greek symbols with subscripts and accents (``x_1``,
``\\bar{x}``, ``\\alpha_{ij}``); ``+ - * /``, ``\\cdot \\times \\div``, implicit multiplication, powers, factorial;
``\\frac \\dfrac \\tfrac \\binom \\sqrt[n]{}``; absolute value bars, ``\\left ... \\right`` of every bracket
kind, floor / ceiling brackets; named functions with optional power and subscript (``\\sin^2 x``,
``\\log_2 8``, ``\\ln``, ``\\exp``, inverse / hyperbolic trigonometry, ``\\gcd \\lcm \\max \\min``) and
``\\operatorname{...}``; applied function symbols ``f(x, y)``; ``\\sum \\prod`` with limits, ``\\int`` with or
without bounds and its differential, ``\\lim_{x \\to a^+}``, derivatives ``\\frac{d}{dx}`` /
``\\frac{\\partial}{\\partial x}``; matrix environments (``matrix pmatrix bmatrix`` -> Matrix, ``vmatrix`` ->
determinant), transpose ``^T`` / ``^{\\top}`` and inverse ``^{-1}`` of matrices; relations
(``= \\neq < > \\le \\ge``) -> Eq / Ne / Lt ...; ``a \\bmod b``; degrees ``^\\circ``; constants ``\\pi``, ``e``,
``\\infty``, the imaginary unit ``i`` (``complex_i=True``); evaluation bars ``expr|_{x=a}``; comma-separated lists
at top level -> tuple. ``\\text{..}`` / ``\\mathrm{..}`` wrappers are read as their content (a differential
``\\mathrm{d}x`` included).

    latex2sympy(r"\\frac{d}{dx}(x^{2}+x)")        -> Derivative(x**2 + x, x)
    latex2sympy(r"\\sum_{i = 1}^{n} i")            -> Sum(i, (i, 1, n))
    latex2sympy(r"\\int_{a}^{b} \\frac{dt}{t}")     -> Integral(1/t, (t, a, b))
    latex2latex(r"\\frac{d}{dx}(x^{2}+x)")         -> "2 x + 1"
"""
import re
from typing import List, Optional, Tuple

import sympy as sp

GREEK = {"alpha", "beta", "gamma", "delta", "epsilon", "varepsilon", "zeta", "eta", "theta", "vartheta", "iota",
         "kappa", "lambda", "mu", "nu", "xi", "rho", "varrho", "sigma", "varsigma", "tau", "upsilon", "phi", "varphi",
         "chi", "psi", "omega", "Gamma", "Delta", "Theta", "Lambda", "Xi", "Pi", "Sigma", "Upsilon", "Phi", "Psi",
         "Omega"}

FUNCTIONS = {
    "sin": sp.sin, "cos": sp.cos, "tan": sp.tan, "cot": sp.cot, "sec": sp.sec, "csc": sp.csc,
    "arcsin": sp.asin, "arccos": sp.acos, "arctan": sp.atan, "arccot": sp.acot, "arcsec": sp.asec, "arccsc": sp.acsc,
    "asin": sp.asin, "acos": sp.acos, "atan": sp.atan,
    "sinh": sp.sinh, "cosh": sp.cosh, "tanh": sp.tanh, "coth": sp.coth, "sech": sp.sech, "csch": sp.csch,
    "arsinh": sp.asinh, "arcosh": sp.acosh, "artanh": sp.atanh, "arcsinh": sp.asinh, "arccosh": sp.acosh,
    "arctanh": sp.atanh,
    "exp": sp.exp, "ln": sp.log, "log": None, "lg": None, "sgn": sp.sign, "sign": sp.sign, "det": None,
    "gcd": sp.gcd, "lcm": sp.lcm, "max": sp.Max, "min": sp.Min, "floor": sp.floor, "ceil": sp.ceiling,
    "arg": sp.arg, "Re": sp.re, "Im": sp.im, "re": sp.re, "im": sp.im, "abs": sp.Abs, "sqrt": sp.sqrt,
}
MULTI_ARG = {"gcd", "lcm", "max", "min"}
INVERSE = {sp.sin: sp.asin, sp.cos: sp.acos, sp.tan: sp.atan, sp.cot: sp.acot, sp.sec: sp.asec, sp.csc: sp.acsc,
           sp.sinh: sp.asinh, sp.cosh: sp.acosh, sp.tanh: sp.atanh}
WRAPPERS = {"text", "mathrm", "mathbf", "mathit", "mathsf", "mathtt", "textbf", "textit", "textrm", "mbox", "bm",
            "boldsymbol", "mathbb", "mathcal", "displaystyle", "textstyle", "rm", "operatorname"}
ACCENTS = {"bar": "bar", "overline": "bar", "hat": "hat", "widehat": "hat", "tilde": "tilde", "widetilde": "tilde",
           "vec": "vec", "dot": "dot", "ddot": "ddot"}
RELATIONS = {"=": sp.Eq, "<": sp.Lt, ">": sp.Gt, "le": sp.Le, "leq": sp.Le, "leqslant": sp.Le, "ge": sp.Ge,
             "geq": sp.Ge, "geqslant": sp.Ge, "ne": sp.Ne, "neq": sp.Ne, "lt": sp.Lt, "gt": sp.Gt, "approx": sp.Eq,
             "equiv": sp.Eq}
SPACING = {"quad", "qquad", "!", ",", ":", ";", " ", "enspace", "thinspace", "medspace", "thickspace", "limits",
           "nolimits", "displaystyle", "textstyle", "scriptstyle", "big", "Big", "bigg", "Bigg", "bigl", "bigr", "Bigl",
           "Bigr", "biggl", "biggr", "Biggl", "Biggr", "mathstrut", "phantom"}
OPEN = {"(": ")", "[": "]", "lbrace": "rbrace", "langle": "rangle", "lfloor": "rfloor", "lceil": "rceil",
        "lvert": "rvert", "|": "|"}


class LatexError(ValueError):
    """The text is not in the supported LaTeX subset."""


_TOKEN = re.compile(r"""
    (?P<num>\d+(?:\.\d+)?|\.\d+)
  | (?P<env>\\(?:begin|end)\s*\{\s*[A-Za-z*]+\s*\})
  | (?P<cmd>\\[A-Za-z]+)
  | (?P<esc>\\[\\{}|%$&#_ ,;:!])
  | (?P<sym>[A-Za-z])
  | (?P<op>[-+*/^_=<>!|(){}\[\],&'%.:;])
  | (?P<ws>\s+|\$)
""", re.X)


def tokenize(text: str) -> List[Tuple[str, str]]:
    """-> [(kind, value)]: 'num', 'cmd' (name without the backslash), 'sym' (one letter), 'op', 'begin' / 'end'
    (environment name), 'row' (the ``\\\\`` of a matrix)."""
    out: List[Tuple[str, str]] = []
    i = 0
    while i < len(text):
        m = _TOKEN.match(text, i)
        if not m:
            raise LatexError(f"cannot read {text[i:i + 10]!r}")
        i = m.end()
        kind = m.lastgroup
        val = m.group(kind)
        if kind == "ws":
            continue
        if kind == "env":
            name = re.search(r"\{\s*([A-Za-z*]+)\s*\}", val).group(1)
            out.append(("begin" if val.startswith("\\begin") else "end", name))
        elif kind == "cmd":
            name = val[1:]
            if name in SPACING:
                continue
            out.append(("cmd", name))
        elif kind == "esc":
            ch = val[1]
            if ch == "\\":
                out.append(("row", "\\\\"))
            elif ch in "{}":
                out.append(("cmd", "lbrace" if ch == "{" else "rbrace"))
            elif ch == "|":
                out.append(("cmd", "Vert"))
            elif ch == "%":
                out.append(("op", "%"))
            # spacing escapes (\, \; \: \! "\ ") and \$ \& \# \_ carry no mathematical meaning here
        else:
            out.append((kind, val))
    return out


class _Parser:
    def __init__(self, tokens, complex_i: bool = False, variable_values=None):
        self.t = tokens
        self.i = 0
        self.complex_i = complex_i
        self.values = variable_values or {}
        self.abs_depth = 0

    # ------------------------------------------------------------------ token helpers
    def peek(self, k: int = 0):
        j = self.i + k
        return self.t[j] if j < len(self.t) else ("eof", "")

    def next(self):
        tok = self.peek()
        self.i += 1
        return tok

    def at(self, kind, val=None, k: int = 0) -> bool:
        a, b = self.peek(k)
        return a == kind and (val is None or b == val)

    def accept(self, kind, val=None) -> bool:
        if self.at(kind, val):
            self.i += 1
            return True
        return False

    def expect(self, kind, val=None):
        if not self.accept(kind, val):
            raise LatexError(f"expected {val or kind}, found {self.peek()[1] or 'end of input'!r}")

    # ------------------------------------------------------------------------ grammar
    def parse(self):
        items = [self.relation()]
        while self.accept("op", ",") or self.accept("op", ";"):
            if self.at("eof"):
                break
            items.append(self.relation())
        if not self.at("eof"):
            raise LatexError(f"unexpected {self.peek()[1]!r}")
        return items[0] if len(items) == 1 else tuple(items)

    def _relop(self):
        kind, val = self.peek()
        if kind == "op" and val in "=<>":
            if val in "<>" and self.at("op", "=", 1):
                self.i += 2
                return sp.Le if val == "<" else sp.Ge
            self.i += 1
            return RELATIONS[val]
        if kind == "cmd" and val in RELATIONS:
            self.i += 1
            return RELATIONS[val]
        return None

    def relation(self):
        lhs = self.additive()
        rels = []
        while True:
            op = self._relop()
            if op is None:
                break
            rhs = self.additive()
            rels.append(op(lhs, rhs, evaluate=False) if op in (sp.Eq, sp.Ne) else op(lhs, rhs))
            lhs = rhs
        if not rels:
            return lhs
        return rels[0] if len(rels) == 1 else sp.And(*rels)

    def additive(self):
        if self.accept("op", "-"):
            left = -self.term()
        elif self.accept("cmd", "pm") or self.accept("cmd", "mp"):
            left = self.term()
        else:
            self.accept("op", "+")
            left = self.term()
        while True:
            if self.accept("op", "+"):
                left = left + self.term()
            elif self.accept("op", "-"):
                left = left - self.term()
            elif self.at("cmd", "pm") or self.at("cmd", "mp"):
                # a +- b: both branches, as a two-element set would lose ordering; keep the '+' branch like graders do
                self.next()
                left = left + self.term()
            else:
                return left

    def _starts_factor(self) -> bool:
        kind, val = self.peek()
        if kind in ("num", "sym", "begin"):
            return True
        if kind == "op":
            if val == "|":
                return self.abs_depth == 0 or not self._bar_closes()
            return val in "({["
        if kind == "cmd":
            if val in ("right", "rbrace", "rangle", "rfloor", "rceil", "rvert", "to", "rightarrow", "pm", "mp", "bmod",
                       "mod", "pmod", "cdot", "times", "div", "cdots", "ldots", "dots", "circ", "degree", "top", "Vert",
                       "mid", "over", "choose", "in", "cup", "cap", "land", "lor", "wedge", "vee", "ast", "implies",
                       "Rightarrow", "iff", "text", "longrightarrow", "Longrightarrow"):
                return val == "text"
            return val not in RELATIONS
        return False

    def _bar_closes(self) -> bool:
        """Inside |...|: a bar closes unless what precedes it demands an operand."""
        pk, pv = self.t[self.i - 1] if self.i > 0 else ("eof", "")
        if pk == "op" and pv in "+-*/^_(,=<>[{":
            return False
        if pk == "cmd" and pv in ("cdot", "times", "div", "left"):
            return False
        return True

    def term(self):
        left = self.unary()
        while True:
            if self.accept("op", "*") or self.accept("cmd", "cdot") or self.accept("cmd", "times") or self.accept(
                    "cmd", "ast"):
                left = left * self.unary()
            elif self.accept("op", "/") or self.accept("cmd", "div") or self.accept("op", ":"):
                left = left / self.unary()
            elif self.accept("cmd", "bmod") or self.accept("cmd", "mod"):
                left = sp.Mod(left, self.unary())
            elif self.at("cmd", "pmod"):
                self.next()
                left = sp.Mod(left, self.group_or_atom())
            elif self._starts_factor():
                left = left * self.power()  # implicit multiplication binds like '*'
            else:
                return left

    def unary(self):
        if self.accept("op", "-"):
            return -self.unary()
        if self.accept("op", "+"):
            return self.unary()
        return self.power()

    def power(self):
        base = self.postfix()
        while True:
            if self.at("op", "^"):
                self.next()
                if self.accept("cmd", "circ") or self.accept("cmd", "degree"):
                    base = base * sp.pi / 180
                    continue
                if self.at("op", "{") and self.at("cmd", "circ", 1) and self.at("op", "}", 2):
                    self.i += 3
                    base = base * sp.pi / 180
                    continue
                exp = self.script()
                if isinstance(base, sp.MatrixBase):
                    if exp in (sp.Symbol("T"), sp.Symbol("top")):
                        base = base.T
                        continue
                    if exp == -1:
                        base = base.inv()
                        continue
                base = _power(base, exp)
            elif self.at("op", "'") and isinstance(base, sp.MatrixBase):
                self.next()
                base = base.T
            else:
                return base

    def script(self):
        """Operand of ^ or _: a braced group or a single token."""
        if self.at("op", "{"):
            return self.braced()
        kind, val = self.peek()
        if kind == "num":
            self.next()
            # x^23 means x^2 * 3 in LaTeX: only the first digit is the exponent
            if len(val) > 1 and "." not in val:
                self.t.insert(self.i, ("num", val[1:]))
                return sp.Integer(val[0])
            return _number(val)
        if kind == "op" and val == "-":
            self.next()
            return -self.script()
        if kind == "cmd" and val in ("top", "prime"):
            self.next()
            return sp.Symbol("top" if val == "top" else "prime")
        if kind == "sym":  # one letter only: in \int^b_a the _a belongs to the integral, not to b
            self.next()
            if val in self.values:
                return sp.sympify(self.values[val])
            return sp.E if val == "e" else sp.I if val in ("i", "I") and self.complex_i else sp.Symbol(val)
        return self.atom()

    def braced(self):
        self.expect("op", "{")
        if self.accept("op", "}"):
            return sp.Integer(1)
        saved = self.abs_depth
        self.abs_depth = 0
        e = self.relation()
        while self.accept("op", ","):
            e = (e if isinstance(e, tuple) else (e,)) + (self.relation(),)
        self.abs_depth = saved
        self.expect("op", "}")
        return e

    def postfix(self):
        e = self.atom()
        while True:
            if self.at("op", "!"):
                self.next()
                # 10^9 ! would be computed digit by digit: numbers beyond a few thousand stay symbolic
                big = isinstance(e, sp.Integer) and abs(int(e)) > 5000
                e = sp.factorial(e, evaluate=False) if big else sp.factorial(e)
            elif self.at("op", "%"):
                self.next()
                e = e / 100
            elif self.at("cmd", "degree"):
                self.next()
                e = e * sp.pi / 180
            elif self.at("op", "|") and self.at("op", "_", 1) and self.abs_depth == 0:
                self.i += 2
                e = self._eval_at(e)
            else:
                return e

    def _eval_at(self, e):
        sub = self.braced() if self.at("op", "{") else self.atom()
        subs = sub if isinstance(sub, tuple) else (sub,)
        for s in subs:
            if not isinstance(s, sp.Eq):
                raise LatexError("evaluation bar needs x=a")
            e = e.subs(s.lhs, s.rhs)
        return e

    # -------------------------------------------------------------------------- atoms
    def group_or_atom(self):
        if self.at("op", "{"):
            return self.braced()
        return self.atom()

    def atom(self):
        kind, val = self.peek()
        if kind == "num":
            self.next()
            return _number(val)
        if kind == "sym":
            return self.symbol()
        if kind == "begin":
            return self.environment()
        if kind == "op":
            if val == "{":
                return self.braced()
            if val in "([":
                return self.bracketed(val)
            if val == "|":
                return self.bars()
        if kind == "cmd":
            return self.command()
        raise LatexError(f"unexpected {val or 'end of input'!r}")

    def bracketed(self, opener: str):
        self.next()
        saved = self.abs_depth
        self.abs_depth = 0
        items = [self.relation()]
        while self.accept("op", ",") or self.accept("op", ";"):
            items.append(self.relation())
        self.abs_depth = saved
        self.accept("cmd", "right")
        kind, val = self.next()
        if not (kind == "op" and val in ")]"):
            raise LatexError(f"unbalanced {opener!r}")
        if len(items) == 1:
            return items[0]
        if len(items) == 2 and (opener, val) != ("(", ")"):
            lo, hi = items  # [a, b) and friends
            try:
                return sp.Interval(lo, hi, left_open=opener == "(", right_open=val == ")")
            except Exception:
                return tuple(items)
        return tuple(items)

    def bars(self):
        self.next()
        self.abs_depth += 1
        e = self.additive()
        self.abs_depth -= 1
        self.accept("cmd", "right")
        self.expect("op", "|")
        return e.det() if isinstance(e, sp.MatrixBase) else sp.Abs(e)

    def symbol(self):
        _, name = self.next()
        # differential handled by the integral; plain letters otherwise
        if name == "e" and not self.at("op", "_"):
            return sp.E
        if name in ("i", "I") and self.complex_i and not self.at("op", "_"):
            return sp.I
        name = self._with_subscript(name)
        return self._apply_if_called(name)

    def _with_subscript(self, name: str) -> str:
        if self.at("op", "_"):
            self.next()
            if self.at("op", "{"):
                depth, j = 0, self.i
                parts = []
                while j < len(self.t):
                    k, v = self.t[j]
                    if k == "op" and v == "{":
                        depth += 1
                        if depth > 1:
                            parts.append(v)
                    elif k == "op" and v == "}":
                        depth -= 1
                        if depth == 0:
                            break
                        parts.append(v)
                    else:
                        parts.append(v)
                    j += 1
                if depth != 0:
                    raise LatexError("unbalanced subscript")
                self.i = j + 1
                sub = "".join(parts)
            else:
                sub = self.next()[1]
            name = f"{name}_{{{sub}}}" if len(sub) > 1 else f"{name}_{sub}"
        return name

    def _apply_if_called(self, name: str):
        if name in self.values:
            return sp.sympify(self.values[name])
        base = name.split("_")[0]
        # f(x), g(x, y): a lone f / g / h (optionally subscripted) followed by a parenthesis is an application
        if base in ("f", "g", "h", "F", "G", "H") and (self.at("op", "(") or (self.at("cmd", "left") and self.at("op", "(", 1))):
            self.accept("cmd", "left")
            args = self.bracketed("(")
            args = args if isinstance(args, tuple) else (args,)
            return sp.Function(name)(*args)
        return sp.Symbol(name)

    def command(self):
        _, name = self.next()
        if name in ("frac", "dfrac", "tfrac", "cfrac"):
            return self.fraction()
        if name in ("binom", "dbinom", "tbinom"):
            n, k = self.group_or_atom(), self.group_or_atom()
            return sp.binomial(n, k)
        if name == "sqrt":
            root = None
            if self.accept("op", "["):
                root = self.additive()
                self.expect("op", "]")
            arg = self.group_or_atom()
            return sp.sqrt(arg) if root is None else sp.Pow(arg, sp.Integer(1) / root)
        if name == "left":
            return self.left_right()
        if name in OPEN:
            return self.delimited(name)
        if name == "Vert":
            e = self.additive()
            self.accept("cmd", "right")
            self.expect("cmd", "Vert")
            return e.norm() if isinstance(e, sp.MatrixBase) else sp.Abs(e)
        if name in ("pi", "Pi") and not self.at("op", "_"):
            return sp.pi if name == "pi" else sp.Symbol("Pi")
        if name == "infty":
            return sp.oo
        if name in ("emptyset", "varnothing"):
            return sp.EmptySet
        if name == "imath":
            return sp.I
        if name in GREEK:
            return self._apply_if_called(self._with_subscript(name))
        if name in ACCENTS:
            inner = self.group_or_atom()
            return sp.Symbol(self._with_subscript(f"{inner}{ACCENTS[name]}"))
        if name in WRAPPERS:
            return self.wrapper(name)
        if name in ("sum", "prod"):
            return self.big_operator(name)
        if name in ("int", "iint", "oint"):
            return self.integral()
        if name == "lim":
            return self.limit()
        if name in FUNCTIONS:
            return self.function(name)
        if name in ("ldots", "cdots", "dots"):
            return sp.Symbol("dots")
        if name == "ell":
            return sp.Symbol("ell")
        if name == "hbar":
            return sp.Symbol("hbar")
        if name in ("circ", "degree"):
            return sp.pi / 180
        raise LatexError(f"unsupported command \\{name}")

    def wrapper(self, name: str):
        if not self.at("op", "{"):
            return self.atom()
        # \text{...}: words are one symbol, math inside is parsed
        j, depth = self.i, 0
        while j < len(self.t):
            k, v = self.t[j]
            if k == "op" and v == "{":
                depth += 1
            elif k == "op" and v == "}":
                depth -= 1
                if depth == 0:
                    break
            j += 1
        inner = self.t[self.i + 1:j]
        if name == "operatorname" or (inner and all(k == "sym" for k, _ in inner)):
            word = "".join(v for _, v in inner)
            if word in FUNCTIONS and len(word) > 1:
                self.i = j + 1
                return self.function(word)
            if len(inner) > 1 and name in ("text", "mbox", "textrm", "textbf", "textit", "operatorname"):
                self.i = j + 1
                return sp.Symbol(word)
        return self.braced()

    def fraction(self):
        # derivative operator \frac{d}{dx} / \frac{d^2}{dx^2} / \frac{\partial}{\partial x}
        save = self.i
        d = self._derivative_head()
        if d is not None:
            var, order, numer = d
            target = numer if numer is not None else self.power_operand()
            return sp.Derivative(target, (var, order)) if order != 1 else sp.Derivative(target, var)
        self.i = save
        num = self.group_or_atom()
        den = self.group_or_atom()
        return num / den

    def power_operand(self):
        """What a prefix operator (derivative, limit, sum) applies to: one multiplicative term."""
        return self.term()

    def _d_token(self) -> bool:
        if self.accept("sym", "d") or self.accept("cmd", "partial"):
            return True
        if self.at("cmd", "mathrm") and self.at("op", "{", 1) and self.at("sym", "d", 2) and self.at("op", "}", 3):
            self.i += 4
            return True
        return False

    def _derivative_head(self):
        """At '{' of \\frac's numerator: parse {d [^n] [f]}{d x [^n]} -> (var, order, numerator expr or None)."""
        if not self.accept("op", "{"):
            return None
        if not self._d_token():
            return None
        order = 1
        if self.accept("op", "^"):
            o = self.script()
            if not (isinstance(o, sp.Integer) and o > 0):
                return None
            order = int(o)
        numer = None
        if not self.at("op", "}"):
            try:
                numer = self.additive()
            except LatexError:
                return None
        if not self.accept("op", "}"):
            return None
        if not self.accept("op", "{"):
            return None
        if not self._d_token():
            return None
        kind, val = self.peek()
        if kind == "sym":
            self.next()
            var = sp.Symbol(self._with_subscript(val))
        elif kind == "cmd" and val in GREEK:
            self.next()
            var = sp.Symbol(self._with_subscript(val))
        else:
            return None
        if self.accept("op", "^"):
            o = self.script()
            if o != order:
                return None
        if not self.accept("op", "}"):
            return None
        return var, order, numer

    def left_right(self):
        kind, val = self.next()
        if kind == "op" and val in "([":
            self.i -= 1
            return self.bracketed(val)
        if kind == "op" and val == "|":
            self.i -= 1
            return self.bars()
        if kind == "op" and val == ".":
            e = self.additive()
            self.accept("cmd", "right")
            if self.at("op", "|") and self.at("op", "_", 1):
                self.i += 2
                return self._eval_at(e)
            self.next()
            return e
        if kind == "cmd" and val in OPEN:
            return self.delimited(val)
        if kind == "cmd" and val == "Vert":
            self.i -= 1
            self.t[self.i] = ("cmd", "Vert")
            return self.command()
        raise LatexError(f"unsupported delimiter after \\left: {val!r}")

    def delimited(self, opener: str):
        closer = OPEN[opener]
        saved = self.abs_depth
        self.abs_depth = 0
        items = [self.relation()]
        while self.accept("op", ","):
            items.append(self.relation())
        self.abs_depth = saved
        self.accept("cmd", "right")
        self.expect("cmd", closer)
        e = items[0] if len(items) == 1 else tuple(items)
        if opener == "lfloor":
            return sp.floor(e)
        if opener == "lceil":
            return sp.ceiling(e)
        if opener == "lvert":
            return e.det() if isinstance(e, sp.MatrixBase) else sp.Abs(e)
        if opener == "lbrace":  # \{a, b\} is a set; \{a + b\} is used as a third kind of parenthesis
            return sp.FiniteSet(*items) if len(items) > 1 else e
        return e

    def function(self, name: str):
        sub = None
        power = None
        for _ in range(2):  # \log_2^3 x or \sin^2_..: either order
            if self.at("op", "_") and sub is None:
                self.next()
                sub = self.script()
            elif self.at("op", "^") and power is None:
                self.next()
                power = self.script()
        if name in MULTI_ARG or self.at("op", "(") or (self.at("cmd", "left") and self.at("op", "(", 1)):
            self.accept("cmd", "left")
            if self.at("op", "("):
                arg = self.bracketed("(")
            else:
                arg = self.braced() if self.at("op", "{") else self.power_operand()
        elif self.at("op", "{"):
            arg = self.braced()
            # \sin{x}y is sin(x)*y, but \sin{2}\pi x is uncommon enough to read the same way
        else:
            arg = self.function_operand()
        args = arg if isinstance(arg, tuple) else (arg,)
        if name == "log":
            val = sp.log(args[0], sub) if sub is not None else sp.log(args[0], 10) if _LOG_BASE10[0] else sp.log(args[0])
        elif name == "lg":
            val = sp.log(args[0], 10)
        elif name == "det":
            val = args[0].det() if isinstance(args[0], sp.MatrixBase) else sp.Function("det")(*args)
        else:
            fn = FUNCTIONS[name]
            if power == -1 and fn in INVERSE:
                fn, power = INVERSE[fn], None
            val = fn(*args)
        return val if power is None else sp.Pow(val, power)

    def function_operand(self):
        """Unparenthesised argument: \\sin 2x, \\ln x^2, \\cos\\theta - a product of factors up to the next
        additive operator or function name."""
        e = self.unary_factor()
        while True:
            kind, val = self.peek()
            if kind == "cmd" and (val in FUNCTIONS or val in ("sum", "prod", "int", "lim")):
                return e
            if self._starts_factor() and not (kind == "op" and val in "(["):
                e = e * self.power()
            else:
                return e

    def unary_factor(self):
        if self.accept("op", "-"):
            return -self.power()
        return self.power()

    def _limits(self):
        lo = hi = None
        for _ in range(2):
            if self.at("op", "_") and lo is None:
                self.next()
                lo = self._limit_operand()
            elif self.at("op", "^") and hi is None:
                self.next()
                hi = self._limit_operand()
        return (None if lo is _EMPTY else lo), (None if hi is _EMPTY else hi)

    def _limit_operand(self):
        if self.at("op", "{") and self.at("op", "}", 1):
            self.i += 2
            return _EMPTY
        return self.braced() if self.at("op", "{") else self.script()

    def big_operator(self, name: str):
        lo, hi = self._limits()
        body = self.power_operand()
        cls = sp.Sum if name == "sum" else sp.Product
        if isinstance(lo, sp.Eq):
            if hi is None:
                raise LatexError(f"\\{name} needs an upper limit")
            return cls(body, (lo.lhs, lo.rhs, hi))
        if lo is None and hi is None:
            raise LatexError(f"\\{name} needs limits")
        raise LatexError(f"\\{name}: lower limit must be i=a")

    def integral(self):
        lo, hi = self._limits()
        # find the differential: scan the factors of the integrand for d<var>
        start = self.i
        depth = 0
        j = start
        dpos = None
        while j < len(self.t):
            k, v = self.t[j]
            if k == "op" and v in "({[":
                depth += 1
            elif k == "op" and v in ")}]":
                if depth == 0:
                    break
                depth -= 1
            elif depth == 0 and k == "op" and v in "=,<>":
                break
            is_d = (k == "sym" and v == "d") or (
                k == "cmd" and v == "mathrm" and self.t[j + 1:j + 4] == [("op", "{"), ("sym", "d"), ("op", "}")])
            if is_d:
                off = 4 if k == "cmd" else 1
                nk, nv = self.t[j + off] if j + off < len(self.t) else ("eof", "")
                if nk == "sym" or (nk == "cmd" and nv in GREEK):
                    dpos = (j, off, depth)
                    if depth == 0:
                        break
            j += 1
        if dpos is None:
            body = self.additive()
            free = sorted(body.free_symbols, key=str)
            var = sp.Symbol("x") if sp.Symbol("x") in free or not free else free[0]
        else:
            j, off, d = dpos
            nk, nv = self.t[j + off]
            # cut the differential out of the token stream, then parse what is left as the integrand
            end = j + off + 1
            var_name = nv
            if end < len(self.t) and self.t[end] == ("op", "_"):
                sub = self.t[end + 1][1]
                var_name, end = f"{nv}_{sub}", end + 2
            var = sp.Symbol(var_name)
            del self.t[j:end]
            if d > 0:
                # \frac{dt}{t}: the differential sat inside a group - leave a 1 in its place
                self.t.insert(j, ("num", "1"))
                body = self.term()
            elif j == start:
                body = sp.Integer(1) if not self._starts_factor() else self.term()
            else:
                sub = _Parser(self.t[start:j], self.complex_i, self.values)
                body = sub.additive()
                if not sub.at("eof"):
                    raise LatexError("cannot read the integrand")
                self.i = j
        if lo is not None and hi is not None:
            return sp.Integral(body, (var, lo, hi))
        return sp.Integral(body, var)

    def limit(self):
        self.expect("op", "_")
        self.expect("op", "{")
        kind, val = self.next()
        if kind == "sym":
            var = sp.Symbol(self._with_subscript(val))
        elif kind == "cmd" and val in GREEK:
            var = sp.Symbol(self._with_subscript(val))
        else:
            raise LatexError("limit variable expected")
        if not any(self.accept("cmd", a) for a in ("to", "rightarrow", "Rightarrow", "longrightarrow", "Longrightarrow")):
            raise LatexError("\\to expected in limit")
        point = self.additive_no_dir()
        direction = "+-"
        if self.accept("op", "^"):
            if self.accept("op", "{"):
                sign = self.next()[1]
                self.expect("op", "}")
            else:
                sign = self.next()[1]
            if sign not in "+-":
                raise LatexError("limit direction must be + or -")
            direction = sign
        self.expect("op", "}")
        body = self.power_operand()
        return sp.Limit(body, var, point, direction)

    def additive_no_dir(self):
        """The limit point: an additive expression in which a trailing ^+ / ^- is NOT an exponent."""
        j, depth = self.i, 0
        while j < len(self.t):
            k, v = self.t[j]
            if k == "op" and v == "{":
                depth += 1
            elif k == "op" and v == "}":
                if depth == 0:
                    break
                depth -= 1
            j += 1
        end = j
        if end - 2 >= self.i and self.t[end - 2] == ("op", "^") and self.t[end - 1][1] in "+-":
            end -= 2
        elif (end - 4 >= self.i and self.t[end - 4] == ("op", "^") and self.t[end - 3] == ("op", "{")
              and self.t[end - 2][1] in "+-" and self.t[end - 1] == ("op", "}")):
            end -= 4
        sub = _Parser(self.t[self.i:end], self.complex_i, self.values)
        e = sub.additive()
        if not sub.at("eof"):
            raise LatexError("cannot read the limit point")
        self.i = end
        return e

    def environment(self):
        _, name = self.next()
        base = name.rstrip("*")
        if base == "array" and self.at("op", "{"):  # column spec
            depth = 0
            while True:
                k, v = self.next()
                if k == "op" and v == "{":
                    depth += 1
                elif k == "op" and v == "}":
                    depth -= 1
                    if depth == 0:
                        break
                elif k == "eof":
                    raise LatexError("unterminated array column spec")
        if base not in ("matrix", "pmatrix", "bmatrix", "Bmatrix", "vmatrix", "Vmatrix", "smallmatrix", "array",
                        "cases", "aligned", "align", "equation", "gathered"):
            raise LatexError(f"unsupported environment {name}")
        rows, row = [], []
        saved = self.abs_depth
        self.abs_depth = 0
        while True:
            if self.at("end"):
                break
            if self.at("eof"):
                raise LatexError(f"unterminated environment {name}")
            if self.accept("row"):
                rows.append(row)
                row = []
                continue
            if self.accept("op", "&"):
                continue
            row.append(self.relation())
            if not (self.at("op", "&") or self.at("row") or self.at("end")):
                # cases: "x, & condition" style trailing text
                if self.accept("op", ","):
                    continue
                raise LatexError(f"unexpected {self.peek()[1]!r} in {name}")
        self.abs_depth = saved
        if row:
            rows.append(row)
        _, closing = self.next()
        if closing != name:
            raise LatexError(f"\\begin{{{name}}} closed by \\end{{{closing}}}")
        if base in ("cases", "aligned", "align", "equation", "gathered"):
            flat = [c for r in rows for c in r]
            return flat[0] if len(flat) == 1 else tuple(flat)
        width = max(len(r) for r in rows) if rows else 0
        if not rows or any(len(r) != width for r in rows):
            raise LatexError("ragged matrix")
        m = sp.Matrix(rows)
        return m.det() if base in ("vmatrix",) else m


def _power(base, exp):
    """base ** exp; an integer tower such as 9^{9^{9}} stays unevaluated instead of being expanded digit by digit."""
    if isinstance(exp, tuple) or isinstance(base, tuple):
        raise LatexError("a list cannot be an operand of ^")
    if (isinstance(base, (sp.Integer, sp.Rational)) and isinstance(exp, (sp.Integer, sp.Rational))
            and abs(exp) > 4096 and base not in (0, 1, -1)):
        return sp.Pow(base, exp, evaluate=False)
    return sp.Pow(base, exp)


_LOG_BASE10 = [True]
_EMPTY = object()


def _number(text: str):
    return sp.Integer(text) if text.isdigit() else sp.Rational(text)


def _preprocess(text: str) -> str:
    text = text.strip()
    for a, b in (("\\dfrac", "\\frac"), ("\\tfrac", "\\frac"), ("\\left\\{", "\\left\\lbrace "),
                 ("\\right\\}", "\\right\\rbrace "), ("\\lvert", "|"), ("\\rvert", "|"), ("\\vert", "|"),
                 ("\\mid", "|"), ("\\colon", ":"), ("\\ge ", "\\geq "), ("\\le ", "\\leq "), ("\\dots", "\\ldots"),
                 ("\\!", ""), ("\\,", " "), ("\\;", " "), ("\\:", " "), ("^{\\prime}", "'"), ("\\%", "%"),
                 ("\\$", ""), ("°", "^\\circ "), ("π", "\\pi "), ("∞", "\\infty "), ("×", "\\times "), ("÷", "\\div "),
                 ("−", "-"), ("·", "\\cdot "), ("≤", "\\leq "), ("≥", "\\geq "), ("≠", "\\neq "), ("√", "\\sqrt ")):
        text = text.replace(a, b)
    text = re.sub(r"^\\\(|\\\)$|^\\\[|\\\]$", "", text).strip()
    # thousands separators inside numbers: 1{,}000 and 1,000 (exactly three digits after each comma)
    text = text.replace("{,}", "").replace("!=", "\\neq ")
    text = re.sub(r"(\d)\s*E\s*(-?\d+)", r"\1\\times 10^{\2}", text)  # 2.5E2
    text = re.sub(r"(?<![\d.])(\d{1,3}(?:,\d{3})+)(?![\d])", lambda m: m.group(1).replace(",", ""), text)
    return text


def latex2sympy(text: str, variable_values: Optional[dict] = None, complex_i: bool = False, log_base_10: bool = True):
    """Parse a LaTeX math string -> a SymPy object (an expression, a relation, a Matrix, an Interval / FiniteSet, or a
    tuple for comma-separated lists). Raises ``LatexError`` on text outside the supported subset. Nothing is
    simplified: sums, integrals, limits and derivatives stay unevaluated (``.doit()`` evaluates them)."""
    import warnings

    _LOG_BASE10[0] = log_base_10
    if len(text) > 4000:
        raise LatexError("expression too long")
    tokens = tokenize(_preprocess(text))
    if not tokens:
        raise LatexError("empty expression")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")  # SymPy's deprecation notes on lists inside matrices / powers
        try:
            return _Parser(tokens, complex_i=complex_i, variable_values=variable_values).parse()
        except RecursionError:
            raise LatexError("expression nested too deeply") from None


def latex2latex(text: str, **kw) -> str:
    """Parse, evaluate (``doit`` + ``simplify``) and print back as LaTeX."""
    e = latex2sympy(text, **kw)
    if isinstance(e, tuple):
        return ", ".join(sp.latex(sp.simplify(x.doit())) for x in e)
    return sp.latex(sp.simplify(e.doit()) if not isinstance(e, sp.MatrixBase) else e.doit())


def try_latex2sympy(text: str, **kw):
    """``latex2sympy`` that returns None instead of raising (grading code tries several readings of an answer)."""
    try:
        return latex2sympy(text, **kw)
    except (LatexError, sp.SympifyError, TypeError, ValueError, AttributeError, IndexError, ZeroDivisionError,
            RecursionError, NotImplementedError):
        return None
