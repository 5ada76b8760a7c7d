threads = False
