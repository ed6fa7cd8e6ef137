"""placeholder -- replaced by the ctypes/autograd bridge"""
