def pretty_print(x):
    return str(x)
