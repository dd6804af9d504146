"""Network factories of the reference's experiment drivers (the shapes, hyper-parameters and state_dict names the
sampling path must accept): experiments/kolmogorov/utils.py:25-81 and experiments/lorenz/utils.py:22-79."""
