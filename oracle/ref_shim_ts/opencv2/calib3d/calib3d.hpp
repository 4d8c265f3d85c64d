#include <opencv2/cv_stub.hpp>
