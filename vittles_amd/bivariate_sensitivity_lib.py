"""Namespace parity with ``vittles/bivariate_sensitivity_lib.py``.  The reference marks these
classes "very experimental and untested" (lines 32-33) and they sit outside the linear-response
hot path: the names exist, construction raises."""


class CrossSensitivity:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError('CrossSensitivity is outside the MI355X hot path of this build.')


class OptimumChecker:
    def __init__(self, *args, **kwargs):
        raise NotImplementedError('OptimumChecker is outside the MI355X hot path of this build.')
