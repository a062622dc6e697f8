from .gaussian_curve_model import GaussianCurveModel, Scene, initialize_bezier_curves  # noqa: F401
