"""Import-compatibility stubs for the reference's matplotlib helpers (`isls/plot_utils.py`): visualisation is
out of scope; the names exist so that notebook imports resolve and fail loudly when called."""


def _out_of_scope(name):
    def f(*a, **k):
        raise NotImplementedError(f"isls.plot_utils.{name}: plotting is out of scope of the MI355X hot path build")
    f.__name__ = name
    return f


plot_robot = _out_of_scope("plot_robot")
plot_robot_base = _out_of_scope("plot_robot_base")
plot_car = _out_of_scope("plot_car")
