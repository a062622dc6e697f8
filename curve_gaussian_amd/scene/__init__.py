from .gaussian_curve_model import GaussianCurveModel  # noqa: F401
