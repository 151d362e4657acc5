"""LaTeX -> SymPy parser (`evaluation/latex2sympy.py`) and its use by the answer grader.

The expected values follow the fixture list of the reference's vendored parser
(``evaluation/latex2sympy/tests/all_good_test.py``, read as data: which LaTeX strings must parse, and to which
mathematical object); the comparison here is mathematical (difference simplifies to zero / same structure), not
on the unevaluated expression trees that list spells out. Deliberate differences: ``f(x)`` is a function
application here (the reference reads it as ``f*x``), and the ANTLR runtime is not used.
"""
import pytest
import sympy as sp
from sympy import (Abs, Derivative, E, Eq, I, Integral, Limit, Matrix, Max, Min, Mod, Ne, Product, Sum, Symbol, acos,
                   acosh, asin, asinh, atan, atanh, binomial, ceiling, cos, cosh, csc, factorial, floor, gcd, lcm, log,
                   oo, pi, root, sec, sin, sinh, sqrt, tan, tanh)

from areal_vllm_amd.evaluation.latex2sympy import LatexError, latex2latex, latex2sympy, tokenize, try_latex2sympy

a, b, c, k, n, t, x, y, z = sp.symbols("a b c k n t x y z")
theta = Symbol("theta")

EXPRESSIONS = [
    ("0", 0), ("1", 1), ("-3.14", sp.Rational("-3.14")), ("5-3", 2), ("(-7.13)(1.5)", sp.Rational("-7.13") * sp.Rational("1.5")),
    ("\\left(-7.13\\right)\\left(1.5\\right)", sp.Rational("-7.13") * sp.Rational("1.5")),
    ("x", x), ("2x", 2 * x), ("x^2", x ** 2), ("x^{3 + 1}", x ** 4), ("x^{\\left\\{3 + 1\\right\\}}", x ** 4),
    ("-3y + 2x", 2 * x - 3 * y), ("-c", -c), ("a \\cdot b", a * b), ("a / b", a / b), ("a \\div b", a / b),
    ("a + b - a", b), ("a\\mod b", Mod(a, b)), ("a \\bmod b", Mod(a, b)),
    ("\\sin \\theta", sin(theta)), ("\\sin(\\theta)", sin(theta)), ("\\sin\\left(\\theta\\right)", sin(theta)),
    ("\\sin^{-1} a", asin(a)), ("\\sin a \\cos b", sin(a) * cos(b)), ("\\sin \\cos \\theta", sin(cos(theta))),
    ("\\sin(\\cos \\theta)", sin(cos(theta))), ("\\arcsin(a)", asin(a)), ("\\arccos(a)", acos(a)),
    ("\\arctan(a)", atan(a)), ("\\sinh(a)", sinh(a)), ("\\cosh(a)", cosh(a)), ("\\tanh(a)", tanh(a)),
    ("\\sinh^{-1}(a)", asinh(a)), ("\\cosh^{-1}(a)", acosh(a)), ("\\tanh^{-1}(a)", atanh(a)),
    ("\\arcsinh(a)", asinh(a)), ("\\arsinh(a)", asinh(a)), ("\\operatorname{arcosh}(a)", acosh(a)),
    ("\\operatorname{artanh}(a)", atanh(a)), ("\\operatorname{gcd}(a, b)", gcd(a, b)), ("\\lcm(a,b)", lcm(a, b)),
    ("\\gcd(12, 18)", 6), ("\\operatorname{floor}(a)", floor(a)), ("\\operatorname{ceil}(b)", ceiling(b)),
    ("\\cos^2(x)", cos(x) ** 2), ("\\cos(x)^2", cos(x) ** 2), ("\\max(a, b)", Max(a, b)), ("\\min(a, b)", Min(a, b)),
    ("\\frac{a}{b}", a / b), ("\\frac{a + b}{c}", (a + b) / c), ("\\frac{7}{3}", sp.Rational(7, 3)),
    ("(\\csc x)(\\sec y)", csc(x) * sec(y)),
    ("\\lim_{x \\to 3} a", Limit(a, x, 3, "+-")), ("\\lim_{x \\rightarrow 3} a", Limit(a, x, 3, "+-")),
    ("\\lim_{x \\Longrightarrow 3} a", Limit(a, x, 3, "+-")), ("\\lim_{x \\to 3^{+}} a", Limit(a, x, 3, "+")),
    ("\\lim_{x \\to 3^{-}} a", Limit(a, x, 3, "-")), ("\\lim_{x \\to \\infty} \\frac{1}{x}", Limit(1 / x, x, oo, "+-")),
    ("\\infty", oo), ("\\infty\\%", oo), ("-\\$\\infty", -oo),
    ("\\frac{d}{dx} x", Derivative(x, x)), ("\\frac{d}{dt} x", Derivative(x, t)),
    ("\\frac{d}{dx} [ \\tan x ]", Derivative(tan(x), x)), ("\\frac{d^2}{dx^2} x^3", Derivative(x ** 3, (x, 2))),
    ("\\frac{\\partial}{\\partial x} x^2 y", Derivative(x ** 2 * y, x)),
    ("|x|", Abs(x)), ("\\left|x\\right|", Abs(x)), ("|x||y|", Abs(x) * Abs(y)), ("\\lfloor x\\rfloor", floor(x)),
    ("\\lceil y\\rceil", ceiling(y)), ("\\pi^{|xy|}", pi ** Abs(x * y)), ("\\frac{\\pi}{3}", pi / 3),
    ("\\sin{\\frac{\\pi}{2}}", 1),
    ("\\int x dx", Integral(x, x)), ("\\int x d\\theta", Integral(x, theta)), ("\\int (x^2 - y)dx", Integral(x ** 2 - y, x)),
    ("\\int x + a dx", Integral(x + a, x)), ("\\int da", Integral(1, a)), ("\\int_0^7 dx", Integral(1, (x, 0, 7))),
    ("\\int_a^b x dx", Integral(x, (x, a, b))), ("\\int^b_a x dx", Integral(x, (x, a, b))),
    ("\\int_{a}^{b} x dx", Integral(x, (x, a, b))), ("\\int_{  }^{}x dx", Integral(x, x)),
    ("\\int (x+a)", Integral(x + a, x)), ("\\int a + b + c dx", Integral(a + b + c, x)),
    ("\\int \\frac{dz}{z}", Integral(1 / z, z)), ("\\int \\frac{3 dz}{z}", Integral(3 / z, z)),
    ("\\int \\frac{1}{x} dx", Integral(1 / x, x)), ("\\int \\frac{1}{a} + \\frac{1}{b} dx", Integral(1 / a + 1 / b, x)),
    ("\\int \\frac{3 \\cdot d\\theta}{\\theta}", Integral(3 / theta, theta)),
    ("\\int \\frac{1}{x} + 1 dx", Integral(1 / x + 1, x)), ("\\int \\frac{1}{x} \\mathrm{d}x", Integral(1 / x, x)),
    ("x_0", Symbol("x_0")), ("x_{1}", Symbol("x_1")), ("x_a", Symbol("x_a")), ("x_{12}", Symbol("x_{12}")),
    ("x!", factorial(x)), ("100!", factorial(100)), ("\\theta!", factorial(theta)), ("(x + 1)!", factorial(x + 1)),
    ("(x!)!", factorial(factorial(x))), ("x!!!", factorial(factorial(factorial(x)))), ("5!7!", factorial(5) * factorial(7)),
    ("\\sqrt{x}", sqrt(x)), ("\\sqrt{x + b}", sqrt(x + b)), ("\\sqrt[3]{\\sin x}", root(sin(x), 3)),
    ("\\sqrt[y]{\\sin x}", root(sin(x), y)),
    ("\\sum_{k = 1}^{3} c", Sum(c, (k, 1, 3))), ("\\sum^3_{k = 1} c", Sum(c, (k, 1, 3))),
    ("\\sum_{k = 1}^{10} k^2", Sum(k ** 2, (k, 1, 10))), ("\\sum_{n = 0}^{\\infty} \\frac{1}{n!}", Sum(1 / factorial(n), (n, 0, oo))),
    ("\\prod_{a = b}^{c} x", Product(x, (a, b, c))), ("\\prod^c_{a = b} x", Product(x, (a, b, c))),
    ("\\ln x", log(x)), ("\\ln xy", log(x * y)), ("\\log x", log(x, 10)), ("\\log xy", log(x * y, 10)),
    ("\\log_{2} x", log(x, 2)), ("\\log_2 x", log(x, 2)), ("\\log_{a} x", log(x, a)), ("\\log_{11} x", log(x, 11)),
    ("\\log_{a^2} x", log(x, a ** 2)), ("[x]", x), ("[a + b]", a + b),
    ("2\\overline{x}", 2 * Symbol("xbar")), ("2\\overline{x}_n", 2 * Symbol("xbar_n")), ("2\\bar{x}_n", 2 * Symbol("xbar_n")),
    ("\\sin\\left(\\theta\\right) \\cdot4", 4 * sin(theta)), ("\\ln\\left(\\left[x-\\theta\\right]\\right)", log(x - theta)),
    ("\\ln\\left(\\left\\{x-\\theta\\right\\}\\right)", log(x - theta)), ("\\ln\\left(\\left|x-\\theta\\right|\\right)", log(Abs(x - theta))),
    ("\\frac{1}{2}xy(x+y)", x * y * (x + y) / 2), ("\\frac{1}{2}\\theta(x+y)", theta * (x + y) / 2),
    ("2.5\\times 10^2", 250), ("1,500\\times 10^{-1}", 150), ("2.5E2", 250), ("1,500E-1", 150),
    ("2x2y", 4 * x * y), ("x2", 2 * x), ("\\binom{5}{2}", 10), ("\\binom{n}{k}", binomial(n, k)), ("50\\%", sp.Rational(1, 2)),
    ("90^\\circ", pi / 2), ("45^{\\circ}", pi / 4), ("x^23", 3 * x ** 2), ("\\dfrac{\\sqrt{3}}{2}", sqrt(3) / 2),
    ("\\left(\\frac{1}{2}\\right)^{3}", sp.Rational(1, 8)), ("e^{2}", E ** 2), ("\\exp(x)", sp.exp(x)),
    ("(2x^3 - x + z)|_{x=3}", z + 51), ("1{,}000", 1000), ("1,000,000", 1000000),
]


@pytest.mark.parametrize("latex,expected", EXPRESSIONS, ids=[e[0] for e in EXPRESSIONS])
def test_expression_parses_to_the_expected_object(latex, expected):
    got = latex2sympy(latex)
    expected = sp.sympify(expected)
    if isinstance(expected, (Integral, Sum, Product, Limit, Derivative)):
        assert type(got) is type(expected)
        assert got.limits == expected.limits if hasattr(expected, "limits") and not isinstance(expected, (Limit, Derivative)) else True
        if isinstance(expected, Limit):
            assert got.args[1:] == expected.args[1:]
        if isinstance(expected, Derivative):
            assert got.variable_count == expected.variable_count
        assert sp.simplify(got.args[0] - expected.args[0]) == 0
    elif expected in (oo, -oo):
        assert got == expected
    else:
        assert sp.simplify(got - expected) == 0, (got, expected)


def test_relations_matrices_sets_and_lists():
    assert latex2sympy("a^2 + b^2 = c^2") == Eq(a ** 2 + b ** 2, c ** 2, evaluate=False)
    assert latex2sympy("a^2 + b^2 != 2c^2") == Ne(a ** 2 + b ** 2, 2 * c ** 2, evaluate=False)
    assert latex2sympy("x < y") == sp.StrictLessThan(x, y)
    assert latex2sympy("x \\leq y") == sp.LessThan(x, y)
    assert latex2sympy("x \\geq y") == sp.GreaterThan(x, y)
    assert latex2sympy("1 < x \\le 5") == sp.And(sp.StrictLessThan(1, x), sp.LessThan(x, 5))
    m = Matrix([[1, 2], [3, 4]])
    for env in ("matrix", "pmatrix", "bmatrix"):
        assert latex2sympy(f"\\begin{{{env}}}1&2\\\\3&4\\end{{{env}}}") == m
    assert latex2sympy("\\begin{matrix}x&x^2\\\\\\sqrt{x}&x\\end{matrix}") == Matrix([[x, x ** 2], [sqrt(x), x]])
    assert latex2sympy("\\begin{pmatrix}1&2\\\\3&4\\end{pmatrix}^T") == m.T
    assert latex2sympy("\\begin{pmatrix}1&2\\\\3&4\\end{pmatrix}^{-1}") == m.inv()
    assert latex2sympy("\\begin{vmatrix} x & 0 \\\\ 0 & x \\end{vmatrix}") == x ** 2
    assert latex2sympy("\\theta\\begin{matrix}1\\\\3\\end{matrix} - \\begin{matrix}-1\\\\2\\end{matrix}") == Matrix(
        [theta + 1, 3 * theta - 2])
    assert latex2sympy("\\{1, 2, 3\\}") == sp.FiniteSet(1, 2, 3)
    assert latex2sympy("[1, 2)") == sp.Interval.Ropen(1, 2)
    assert latex2sympy("(1, 2)") == (1, 2)
    assert latex2sympy("x = 1, y = 2") == (Eq(x, 1), Eq(y, 2))
    f = sp.Function("f")
    assert latex2sympy("f(x) = x^2") == Eq(f(x), x ** 2)
    assert latex2sympy("a+bI", complex_i=True) == a + b * I
    assert sp.simplify(latex2sympy("e^{I\\pi}", complex_i=True)) == -1
    assert latex2sympy("3\\text{ cm}") == 3 * Symbol("cm")
    assert latex2sympy("x", variable_values={"x": 3}) == 3


def test_evaluation_round_trip_and_errors():
    assert latex2latex("\\frac{d}{dx}(x^{2}+x)") == "2 x + 1"
    assert latex2latex("\\sum_{i = 1}^{n} i") == "\\frac{n \\left(n + 1\\right)}{2}"
    assert sp.simplify(latex2sympy("\\int_{a}^{b} \\frac{dt}{t}").doit() - (log(b) - log(a))) == 0
    assert latex2sympy("\\lim_{x \\to 0^+} \\frac{\\sin x}{x}").doit() == 1
    for bad in ["", "\\frac{1}", "(", "x +", "\\begin{pmatrix}1&2\\\\3\\end{pmatrix}", "\\unknowncommand{x}", "|x",
                "\\sum_{k}^{3} k", "\\lim_{x} x", "x ^", "\\begin{matrix}1\\end{pmatrix}", "@"]:
        with pytest.raises(LatexError):
            latex2sympy(bad)
        assert try_latex2sympy(bad) is None
    assert tokenize("\\frac{1}{2}\\,x") == [("cmd", "frac"), ("op", "{"), ("num", "1"), ("op", "}"), ("op", "{"),
                                             ("num", "2"), ("op", "}"), ("sym", "x")]


def test_grader_uses_the_latex_grammar():
    from areal_vllm_amd.reward.math_parser import math_equal, symbolic_equal

    assert symbolic_equal("\\int_0^1 x \\, dx", "\\frac{1}{2}")
    assert symbolic_equal("\\sum_{k=1}^{4} k^2", "30")
    assert symbolic_equal("\\log_2 8", "3")
    assert symbolic_equal("\\sin^2 x + \\cos^2 x", "1")
    assert symbolic_equal("\\binom{6}{2}", "15")
    assert symbolic_equal("\\sqrt[3]{27}", "3")
    assert symbolic_equal("\\begin{vmatrix} 1 & 2 \\\\ 3 & 4 \\end{vmatrix}", "-2")
    assert not symbolic_equal("\\log_2 8", "4")
    assert math_equal("\\frac{\\sqrt{3}}{2}", "\\sin\\frac{\\pi}{3}")
    assert math_equal("x^2+2x+1", "(x+1)^2")
    assert not math_equal("\\frac{1}{3}", "\\frac{1}{2}")
