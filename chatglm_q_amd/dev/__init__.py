"""Developer-only experiments (see experiments.py); nothing in the product path imports this package eagerly."""
