"""placeholder, replaced below"""
class SuperSloMo(object):
    pass
