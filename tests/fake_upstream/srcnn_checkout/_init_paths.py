# upstream adds lib/ to sys.path here
