#include <opencv2/opencv.hpp>
