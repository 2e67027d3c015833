"""Level-set shape generators (input side of the hot path; mirrors the public API of torcwa/geometry.py:4-290).

Every primitive returns sigmoid(edge_sharpness * level) on the cell-centre grid x_i = (i+0.5) Lx/nx, where
`level` is 1 - (normalised distance in the shape's own rotated frame).  Differentiable torch elementwise code.
`geometry` is the instance API; `rcwa_geo` is the legacy class-level twin the notebooks use.
"""
import torch


def _rot(x, y, cx, cy, theta, dtype, device):
    th = torch.as_tensor(theta, dtype=dtype, device=device)
    c, s = torch.cos(th), torch.sin(th)
    return (x - cx) * c + (y - cy) * s, -(x - cx) * s + (y - cy) * c


class _Shapes:
    """Shape algebra shared by the instance and the class-level front-ends; `g` supplies Lx, Ly, nx, ny, ... ."""

    @staticmethod
    def _grid(g):
        g.x = (g.Lx / g.nx) * (torch.arange(g.nx, dtype=g.dtype, device=g.device) + 0.5)
        g.y = (g.Ly / g.ny) * (torch.arange(g.ny, dtype=g.dtype, device=g.device) + 0.5)
        g.x_grid, g.y_grid = torch.meshgrid(g.x, g.y, indexing="ij")

    @staticmethod
    def _level(g, kind, a, b, Cx, Cy, theta=0., power=2.):
        _Shapes._grid(g)
        u, v = _rot(g.x_grid, g.y_grid, Cx, Cy, theta, g.dtype, g.device)
        if kind == "ellipse":
            dist = torch.sqrt((u / a) ** 2 + (v / b) ** 2)
        elif kind == "box":
            dist = torch.maximum(torch.abs(u / (a / 2.)), torch.abs(v / (b / 2.)))
        elif kind == "rhombus":
            dist = torch.abs(u / (a / 2.)) + torch.abs(v / (b / 2.))
        else:  # super ellipse
            dist = (torch.abs(u / (a / 2.)) ** power + torch.abs(v / (b / 2.)) ** power) ** (1 / power)
        return torch.sigmoid(g.edge_sharpness * (1. - dist))


class geometry:
    def __init__(self, Lx: float = 1., Ly: float = 1., nx: int = 100, ny: int = 100, edge_sharpness: float = 1000., *,
                 dtype=torch.float32, device=torch.device("cuda" if torch.cuda.is_available() else "cpu")):
        self.Lx, self.Ly, self.nx, self.ny, self.edge_sharpness = Lx, Ly, nx, ny, edge_sharpness
        self.dtype, self.device = dtype, device

    def grid(self):
        _Shapes._grid(self)

    def circle(self, R, Cx, Cy):
        _Shapes._grid(self)
        return torch.sigmoid(self.edge_sharpness * (1. - torch.sqrt(((self.x_grid - Cx) / R) ** 2 + ((self.y_grid - Cy) / R) ** 2)))

    def ellipse(self, Rx, Ry, Cx, Cy, theta=0.):
        return _Shapes._level(self, "ellipse", Rx, Ry, Cx, Cy, theta)

    def square(self, W, Cx, Cy, theta=0.):
        return _Shapes._level(self, "box", W, W, Cx, Cy, theta)

    def rectangle(self, Wx, Wy, Cx, Cy, theta=0.):
        return _Shapes._level(self, "box", Wx, Wy, Cx, Cy, theta)

    def rhombus(self, Wx, Wy, Cx, Cy, theta=0.):
        return _Shapes._level(self, "rhombus", Wx, Wy, Cx, Cy, theta)

    def super_ellipse(self, Wx, Wy, Cx, Cy, theta=0., power=2.):
        return _Shapes._level(self, "super", Wx, Wy, Cx, Cy, theta, power)

    @staticmethod
    def union(A, B):
        return torch.maximum(A, B)

    @staticmethod
    def intersection(A, B):
        return torch.minimum(A, B)

    @staticmethod
    def difference(A, B):
        return torch.minimum(A, 1. - B)


class rcwa_geo:
    """Legacy class-level API of the notebooks: `rcwa_geo.Lx = ...; rcwa_geo.grid(); rcwa_geo.rectangle(...)` on the class itself."""
    dtype = torch.float32
    device = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    Lx, Ly, nx, ny = 1., 1., 100, 100
    edge_sharpness = 1000.

    @classmethod
    def grid(cls):
        _Shapes._grid(cls)

    @classmethod
    def circle(cls, R, Cx, Cy):
        _Shapes._grid(cls)
        return torch.sigmoid(cls.edge_sharpness * (1. - torch.sqrt(((cls.x_grid - Cx) / R) ** 2 + ((cls.y_grid - Cy) / R) ** 2)))

    @classmethod
    def ellipse(cls, Rx, Ry, Cx, Cy, theta=0.):
        return _Shapes._level(cls, "ellipse", Rx, Ry, Cx, Cy, theta)

    @classmethod
    def square(cls, W, Cx, Cy, theta=0.):
        return _Shapes._level(cls, "box", W, W, Cx, Cy, theta)

    @classmethod
    def rectangle(cls, Wx, Wy, Cx, Cy, theta=0.):
        return _Shapes._level(cls, "box", Wx, Wy, Cx, Cy, theta)

    @classmethod
    def rhombus(cls, Wx, Wy, Cx, Cy, theta=0.):
        return _Shapes._level(cls, "rhombus", Wx, Wy, Cx, Cy, theta)

    @classmethod
    def super_ellipse(cls, Wx, Wy, Cx, Cy, theta=0., power=2.):
        return _Shapes._level(cls, "super", Wx, Wy, Cx, Cy, theta, power)

    union = staticmethod(geometry.union)
    intersection = staticmethod(geometry.intersection)
    difference = staticmethod(geometry.difference)
