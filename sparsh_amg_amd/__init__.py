"""placeholder; replaced below once the native library exists"""
