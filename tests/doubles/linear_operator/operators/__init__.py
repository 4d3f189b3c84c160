"""TEST DOUBLE -- linear_operator.operators.LinearOperator's constructor / representation / dispatch contract.

What the real class documents and the reference's operators rely on
(manifold_gp/operators/graph_laplacian_operator.py:35-43):
  * `__init__(*args, **kwargs)`: positional args and tensor / LinearOperator kwargs are the DIFFERENTIABLE
    representation, every other kwarg is kept to re-call the constructor;
  * `representation()`: the flat tuple of tensors (nested operators flattened); anything else in `args` raises;
  * `representation_tree()(*tensors)`: `cls(*args, **kwargs)` with the tensors put back -- how every autograd Function
    of the library (Matmul, Solve, InvQuadLogdet) gets its operator inside `forward`;
  * public `matmul / to_dense / diagonal / mT / shape` dispatch to `_matmul / _diagonal / _transpose_nonbatch / _size`;
  * `_solve(rhs, preconditioner, num_tridiag=0)` returns the solves, or `(solves, tridiagonals)` when num_tridiag > 0.
"""
import itertools
from collections import OrderedDict

import torch


class _RepresentationTree:
    def __init__(self, op):
        self.cls = op.__class__
        self.kwargs = dict(op._nondifferentiable_kwargs)
        self.children = []
        counter = 0
        for name, arg in itertools.chain(((None, a) for a in op._args), op._differentiable_kwargs.items()):
            if isinstance(arg, LinearOperator):
                size = len(arg.representation())
                self.children.append((name, slice(counter, counter + size), arg.representation_tree()))
                counter += size
            else:
                self.children.append((name, counter, None))
                counter += 1

    def __call__(self, *flat):
        args, kwargs = [], {}
        for name, index, subtree in self.children:
            val = subtree(*flat[index]) if subtree is not None else flat[index]
            if name is None:
                args.append(val)
            else:
                kwargs[name] = val
        return self.cls(*args, **kwargs, **self.kwargs)


class LinearOperator:
    def __init__(self, *args, **kwargs):
        self._args = args
        self._differentiable_kwargs = OrderedDict()
        self._nondifferentiable_kwargs = dict()
        for name, val in sorted(kwargs.items()):
            if torch.is_tensor(val) or isinstance(val, LinearOperator):
                self._differentiable_kwargs[name] = val
            else:
                self._nondifferentiable_kwargs[name] = val

    @property
    def _kwargs(self):
        return {**self._differentiable_kwargs, **self._nondifferentiable_kwargs}

    # ---- hooks
    def _matmul(self, rhs):
        raise NotImplementedError

    def _size(self):
        raise NotImplementedError

    def _transpose_nonbatch(self):
        raise NotImplementedError

    def _diagonal(self):
        raise NotImplementedError

    # ---- representation
    def representation(self):
        rep = []
        for arg in itertools.chain(self._args, self._differentiable_kwargs.values()):
            if torch.is_tensor(arg):
                rep.append(arg)
            elif isinstance(arg, LinearOperator):
                rep += list(arg.representation())
            else:
                raise RuntimeError("Representation of a LinearOperator should consist only of Tensors")
        return tuple(rep)

    def representation_tree(self):
        return _RepresentationTree(self)

    # ---- public surface (the library's Functions rebuild the operator from its representation first)
    def _rebuilt(self):
        return self.representation_tree()(*self.representation())

    @property
    def shape(self):
        return self._size()

    def size(self, dim=None):
        s = self._size()
        return s if dim is None else s[dim]

    def dim(self):
        return len(self._size())

    @property
    def dtype(self):
        return self.representation()[0].dtype if self._args and torch.is_tensor(self._args[0]) else torch.float32

    @property
    def device(self):
        return self.representation()[0].device

    def matmul(self, other):
        if other.dim() == 1:
            return self._rebuilt()._matmul(other.unsqueeze(-1)).squeeze(-1)
        return self._rebuilt()._matmul(other)

    __matmul__ = matmul

    @property
    def mT(self):
        return self._transpose_nonbatch()

    T = mT

    def diagonal(self, offset=0, dim1=-2, dim2=-1):
        return self._diagonal()

    def to_dense(self):
        n = self._size()[-1]
        return self.matmul(torch.eye(n, dtype=self.dtype, device=self.device))

    # library drivers of the `_solve` convention (Solve / InvQuadLogdet functions of the real package)
    def solve(self, right_tensor, left_tensor=None):
        sol = self._rebuilt()._solve(right_tensor if right_tensor.dim() > 1 else right_tensor.unsqueeze(-1), None)
        sol = sol if right_tensor.dim() > 1 else sol.squeeze(-1)
        return sol if left_tensor is None else left_tensor @ sol

    def _solve(self, rhs, preconditioner, num_tridiag=0):
        raise NotImplementedError("the double has no linear_cg: operators bring their own _solve")


class _Root(LinearOperator):
    def __init__(self, root):
        super().__init__(root)
        self.root = root

    def _size(self):
        return torch.Size([self.root.shape[-2], self.root.shape[-2]])

    def _matmul(self, rhs):
        return self.root @ (self.root.mT @ rhs)

    def _transpose_nonbatch(self):
        return self

    def to_dense(self):
        return self.root @ self.root.mT


class RootLinearOperator(_Root):
    pass


class LowRankRootLinearOperator(_Root):
    pass


class MatmulLinearOperator(LinearOperator):
    def __init__(self, left, right):
        super().__init__(left, right)
        self.left, self.right = left, right

    def _size(self):
        return torch.Size([self.left.shape[-2], self.right.shape[-1]])

    def _matmul(self, rhs):
        return self.left @ (self.right @ rhs)

    def to_dense(self):
        return self.left @ self.right
