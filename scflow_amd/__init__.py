"""scflow_amd -- MI355X-native implementation of SCFlow's recurrent
flow/pose refinement hot path."""
__version__ = '0.1.0'
