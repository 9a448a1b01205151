from .spring_mass_warp import SpringMassSystemWarp  # noqa: F401  (reference: sim/physics/__init__.py:2 region)
