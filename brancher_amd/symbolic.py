"""
Symbolic link expressions.

The reference stores a link as an opaque Python closure (`brancher/variables.py:977-1002`
keeps only ``fn``; `brancher/functions.py:28-41` closes over torch callables), which is
why every ELBO evaluation has to walk Python.  Here a link is an explicit expression DAG
(``Expr``) that the lowering pass (`lowering.py`) turns into kernel bytecode once.

Node kinds
----------
``var``      leaf: the value of a Variable                       (attr = Variable)
``const``    leaf: python number / numpy array                   (attr = value)
``add sub mul truediv pow``  binary operators                    (`variables.py:995-1002`)
``call``     a named backend function, e.g. ``sigmoid``          (attr = (name|callable, kwargs))
``getitem``  indexing after the sample axis                      (attr = key; `variables.py:279-289,1037-1053`)
``tuple``    tuple of sub-expressions                            (`variables.py:917-919`)
``shape``    the shape of the argument                           (`variables.py:291-295,1055-1061`)
"""
import numbers

import numpy as np

BINARY_OPS = ("add", "sub", "mul", "truediv", "pow")
BINARY_SYMBOLS = {"add": "+", "sub": "-", "mul": "*", "truediv": "/", "pow": "**"}


class Expr:
    __slots__ = ("op", "args", "attr", "_hash")

    def __init__(self, op, args=(), attr=None):
        self.op = op
        self.args = tuple(args)
        self.attr = attr
        self._hash = None

    # structural identity is used for common-subexpression elimination in the lowering
    def key(self):
        if self.op == "var":
            return ("var", id(self.attr))
        if self.op == "const":
            v = self.attr
            if isinstance(v, np.ndarray):
                return ("const", v.shape, v.dtype.str, v.tobytes())
            return ("const", repr(v))
        if self.op == "call":
            fn, kwargs = self.attr
            fkey = fn if isinstance(fn, str) else ("callable", id(fn))
            kw = tuple(sorted((k, v.key() if isinstance(v, Expr) else repr(v)) for k, v in kwargs.items()))
            return ("call", fkey, tuple(a.key() if isinstance(a, Expr) else repr(a) for a in self.args), kw)
        if self.op == "getitem":
            return ("getitem", repr(self.attr), tuple(a.key() for a in self.args))
        return (self.op, tuple(a.key() for a in self.args))

    def variables(self):
        out = set()
        stack = [self]
        while stack:
            e = stack.pop()
            if e.op == "var":
                out.add(e.attr)
            for a in e.args:
                if isinstance(a, Expr):
                    stack.append(a)
            if e.op == "call":
                for v in e.attr[1].values():
                    if isinstance(v, Expr):
                        stack.append(v)
        return out

    def __repr__(self):
        if self.op == "var":
            return "var(%s)" % self.attr.name
        if self.op == "const":
            return "const(%r)" % (self.attr,)
        if self.op == "call":
            fn = self.attr[0]
            return "%s(%s)" % (fn if isinstance(fn, str) else getattr(fn, "__name__", "fn"),
                               ", ".join(repr(a) for a in self.args))
        if self.op in BINARY_SYMBOLS:
            return "(%r %s %r)" % (self.args[0], BINARY_SYMBOLS[self.op], self.args[1])
        return "%s(%s)" % (self.op, ", ".join(repr(a) for a in self.args))


def is_numeric_constant(x):
    return isinstance(x, (numbers.Number, np.ndarray, np.generic))


def const(value):
    return Expr("const", (), value)


def variable(var):
    return Expr("var", (), var)


def binary(op, a, b):
    assert op in BINARY_OPS
    return Expr(op, (a, b))


def call(fn, args, kwargs=None):
    return Expr("call", tuple(args), (fn, dict(kwargs or {})))
